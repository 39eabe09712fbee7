"""Instruction-class sequence per barrier segment of the kernel tools/kstat.py compiled last (/tmp/kstat.s).
M mfma, v valu, p packed valu, t transcendental, a accvgpr move, r LDS read, W LDS write, g vmem, | s_waitcnt, s salu, X scratch.
usage: python tools/kseq.py <kernel-name-substring>"""
import re, sys
txt = open('/tmp/kstat.s').read()
m = re.search(r"^(\S*%s\S*):[^\n]*\n(.*?)s_endpgm" % re.escape(sys.argv[1]), txt, re.S | re.M)
segs = [[]]
for l in m.group(2).splitlines():
    l = l.strip()
    if not l or l[0] in ".;/" or l.endswith(":"): continue
    op = l.split()[0]
    if op == "s_barrier": segs.append([]); continue
    segs[-1].append(op)
def c(op):
    if op.startswith("v_mfma"): return "M"
    if op.startswith(("v_exp", "v_rcp", "v_rsq")): return "t"
    if op.startswith("v_pk"): return "p"
    if op.startswith("v_accvgpr"): return "a"
    if op.startswith("v_"): return "v"
    if op.startswith(("ds_read", "ds_load")): return "r"
    if op.startswith("ds_"): return "W"
    if op.startswith("s_waitcnt"): return "|"
    if op.startswith(("global_", "buffer_", "flat_")): return "g"
    if op.startswith("scratch"): return "X"
    return "s"
for i, s in enumerate(segs):
    print(i, "".join(c(o) for o in s)[:int(sys.argv[2]) if len(sys.argv) > 2 else 600])
