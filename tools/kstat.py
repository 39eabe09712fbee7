"""Static view of one kernel of a csrc/*.hip file: registers / scratch of every kernel matching a pattern and the per-barrier
instruction mix of the first match.   python tools/kstat.py xattn.hip xattn_kernelILi0ELi1ELi1ELi1ELi4ELb0"""
import collections, os, re, subprocess, sys
src, pat = sys.argv[1], sys.argv[2]
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ap-adapter_amd", "csrc")
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17"] + (["-mllvm", "-amdgpu-use-amdgpu-trackers=1"] if src == "mlp.hip" else [] if src in ("mlp3.hip", "geglu3.hip") else ["-mllvm", "-amdgpu-mfma-vgpr-form"]) + sys.argv[3:]
r = subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-S", "--cuda-device-only", src, "-o", "/tmp/kstat.s", "-Rpass-analysis=kernel-resource-usage"],
                   cwd=csrc, capture_output=True, text=True)
if r.returncode:
    print(r.stderr[-3000:]); sys.exit(1)
cur = None
for l in r.stderr.splitlines():
    m = re.search(r"remark: (.*?)(\s*\[-Rpass.*)?$", l)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"): cur = t.split(":", 1)[1].strip()
    elif cur and pat in cur and (t.startswith("VGPRs:") or t.startswith("ScratchSize") or t.startswith("SGPRs:")): print(cur[:90], t)
txt = open("/tmp/kstat.s").read()
m = re.search(r"^(\S*%s\S*):[^\n]*\n(.*?)s_endpgm" % re.escape(pat), txt, re.S | re.M)
if not m: sys.exit("kernel not found")
print("==", m.group(1))
segs = [[]]
for l in m.group(2).splitlines():
    l = l.strip()
    if not l or l[0] in ".;/" or l.endswith(":"): continue
    op = l.split()[0]
    if op == "s_barrier": segs.append([])
    else: segs[-1].append(op)
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_exp", "v_rcp", "v_rsq", "v_log", "v_sqrt")): return "trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "SPILL"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    return "salu"
for i, s in enumerate(segs):
    print(i, len(s), dict(collections.Counter(cls(o) for o in s)))
if "-v" in os.environ.get("KSTAT", ""):
    for i, s in enumerate(segs):
        print(i, collections.Counter(o for o in s if cls(o) in ("valu", "salu")).most_common(30))
