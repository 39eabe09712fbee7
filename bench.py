"""Benchmark of the AP-adapter hot path on MI355X: 10 s clips / second at 200 DDIM steps, AudioLDM2-large + AP.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--la La]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A bench "step" is one DDIM step of the captured hot path over one batch: UNet on the CFG-duplicated batch (2B
sample-forwards), CFG combine, DDIM update -- replayed as one hipGraph.  value = clips/s = (B * N) / (200 * t_step),
t_step = max over ranks of (elapsed / K).  Inputs (latents, GPT-2 / T5 embeddings, audio tokens) are resident in HBM
before the timed region; weights are random-init of the AudioLDM2-large geometry (no checkpoint offline).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DDIM_STEPS_PER_CLIP = 200
SELF_ATTN_KERNEL_SUBSTR = "sattn_fused_kernel<0, 32"  # name of the 1000-token self-attention kernel in the rocprofv3 summaries
DTYPE_NAME = {torch.bfloat16: "bf16", torch.float16: "f16", torch.float32: "f32"}
MFMA_PEAK_TFLOPS = 2500.0   # bf16 dense, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def unet_flops_per_sample(La, t5_len=16):
    """Algorithmic FLOPs (2*MACs) of one UNet sample-forward, AudioLDM2-large geometry at 10 s (NOTES.md 'FLOP model')."""
    boc = (128, 256, 384, 640)
    hw = [250 * 16, 125 * 8, 63 * 4, 32 * 2]
    fl = 0.0
    conv = lambda n, cin, cout, k=9: 2.0 * n * cin * cout * k

    def resnet(n, cin, cout):
        f = conv(n, cin, cout) + conv(n, cout, cout) + 2.0 * 512 * cout
        if cin != cout:
            f += conv(n, cin, cout, 1)
        return f

    def tblock(n, c, kind):
        f = 8.0 * n * c * c + 4.0 * n * n * c                       # attn1 (q,k,v,out + core)
        if kind == "self":
            f += 8.0 * n * c * c + 4.0 * n * n * c
        elif kind == "ip":                                          # K/V hoisted out of the loop
            f += 4.0 * n * c * c + 4.0 * n * (8 + La) * c
        else:
            f += 4.0 * n * c * c + 4.0 * n * t5_len * c
        return f + 24.0 * n * c * c                                 # GEGLU FF

    def layer(n, c):
        f = 0.0
        for kind in ("self", "ip", "t5", "self"):
            f += 4.0 * n * c * c + 2 * tblock(n, c, kind)           # proj_in/out + 2 blocks
        return f

    fl += conv(hw[0], 8, 128)
    # down
    fl += 2 * resnet(hw[0], 128, 128) + conv(hw[1], 128, 128)
    fl += resnet(hw[1], 128, 256) + resnet(hw[1], 256, 256) + 2 * layer(hw[1], 256) + conv(hw[2], 256, 256)
    fl += resnet(hw[2], 256, 384) + resnet(hw[2], 384, 384) + 2 * layer(hw[2], 384) + conv(hw[3], 384, 384)
    fl += resnet(hw[3], 384, 640) + resnet(hw[3], 640, 640) + 2 * layer(hw[3], 640)
    # mid
    fl += 2 * resnet(hw[3], 640, 640) + layer(hw[3], 640)
    # up
    fl += resnet(hw[3], 1280, 640) * 2 + resnet(hw[3], 1024, 640) + 3 * layer(hw[3], 640) + conv(hw[2], 640, 640)
    fl += resnet(hw[2], 1024, 384) + resnet(hw[2], 768, 384) + resnet(hw[2], 640, 384) + 3 * layer(hw[2], 384) + conv(hw[1], 384, 384)
    fl += resnet(hw[1], 640, 256) + resnet(hw[1], 512, 256) + resnet(hw[1], 384, 256) + 3 * layer(hw[1], 256) + conv(hw[0], 256, 256)
    fl += resnet(hw[0], 384, 128) + 2 * resnet(hw[0], 256, 128)
    fl += conv(hw[0], 128, 8)
    return fl


def unet_min_bytes_per_step(B2, La, t5_len=16):
    """Algorithmic HBM bytes of one UNet step under the CURRENT launch structure: the sum over launches of their minimal operand bytes -- every
    launch reads each activation operand once and writes its result once (2 bytes per element, B' sample-forwards), weights once per launch;
    intermediates that stay inside a fused launch (q / k / v, the 4C-wide feed-forward activation at C = 256, scores) do not count.  The
    denominator of hbm_whole_step: measured bytes / this = re-reads, partial-line writes and passes a further fusion would remove."""
    hw = [250 * 16, 125 * 8, 63 * 4, 32 * 2]
    act = [0.0]
    wgt = [0.0]

    def op(n, cin, cout, k=1, res=False, extra_out=0.0):
        act[0] += n * cin + n * cout * (2 if res else 1) + extra_out
        wgt[0] += cin * cout * k

    def norm(n, c):
        act[0] += 2.0 * n * c

    def resnet(n, cin, cout):
        norm(n, cin); op(n, cin, cout, 9); norm(n, cout)
        if cin != cout:
            op(n, cin, cout, 1)
        op(n, cout, cout, 9, res=True)

    def tblock(n, c, kind):
        def self_attn():
            op(n, c, c, 3)            # LayerNorm + q|k|v + attention in one launch (x in, O out)
            op(n, c, c, 1, res=True)  # to_out + residual
        self_attn()
        if kind == "self":
            self_attn()
        else:                         # one-launch cross-attention sub-layer (x in, out; to_q + to_out weights; hoisted K / V)
            keys = (8 + La) if kind == "ip" else t5_len
            op(n, c, c, 2)
            act[0] += 2.0 * keys * c
            if c == 640:              # the 64-token level: attention part + to_out part
                act[0] += 2.0 * n * c
        if c == 256:
            op(n, c, c, 12)           # the whole feed-forward in one launch
        else:
            op(n, c, 4 * c, 2)        # LN + GEGLU projection -> 4C-wide activation
            op(n, 4 * c, c, 1, res=True)

    def layer(n, c):
        for kind in ("self", "ip", "t5", "self"):
            norm(n, c); op(n, c, c); tblock(n, c, kind); tblock(n, c, kind); op(n, c, c, res=True)

    op(hw[0], 8, 128, 9)
    resnet(hw[0], 128, 128); resnet(hw[0], 128, 128); op(hw[1], 128, 128, 9)
    resnet(hw[1], 128, 256); layer(hw[1], 256); resnet(hw[1], 256, 256); layer(hw[1], 256); op(hw[2], 256, 256, 9)
    resnet(hw[2], 256, 384); layer(hw[2], 384); resnet(hw[2], 384, 384); layer(hw[2], 384); op(hw[3], 384, 384, 9)
    resnet(hw[3], 384, 640); layer(hw[3], 640); resnet(hw[3], 640, 640); layer(hw[3], 640)
    resnet(hw[3], 640, 640); layer(hw[3], 640); resnet(hw[3], 640, 640)
    for cin in (1280, 1280, 1024):
        resnet(hw[3], cin, 640); layer(hw[3], 640)
    op(hw[2], 640, 640, 9)
    for cin in (1024, 768, 640):
        resnet(hw[2], cin, 384); layer(hw[2], 384)
    op(hw[1], 384, 384, 9)
    for cin in (640, 512, 384):
        resnet(hw[1], cin, 256); layer(hw[1], 256)
    op(hw[0], 256, 256, 9)
    for cin in (384, 256, 256):
        resnet(hw[0], cin, 128)
    norm(hw[0], 128); op(hw[0], 128, 8, 9)
    return 2.0 * (act[0] * B2 + wgt[0])


def cfg_shared_prefix_flops():
    """FLOPs of the part of a sample-forward that reads no condition (conv_in, the first down block, the first resnet + the
    double-self-attention transformer + proj_in / attn1 of the first conditioned one at the 1000-pixel level): identical rows in both
    halves of the CFG batch, executed once per clip by the shared-prefix step (unet.cfg_expand)"""
    n0, n1 = 250 * 16, 125 * 8
    conv = lambda n, cin, cout, k=9: 2.0 * n * cin * cout * k
    res = lambda n, cin, cout: conv(n, cin, cout) + conv(n, cout, cout) + 2.0 * 512 * cout + (conv(n, cin, cout, 1) if cin != cout else 0.0)
    c = 256
    self_block = 2 * (8.0 * n1 * c * c + 4.0 * n1 * n1 * c) + 24.0 * n1 * c * c
    return (conv(n0, 8, 128) + 2 * res(n0, 128, 128) + conv(n1, 128, 128) + res(n1, 128, 256)
            + 4.0 * n1 * c * c + 2 * self_block                      # transformer 1 (no condition): proj_in/out + 2 blocks
            + 2.0 * n1 * c * c + 8.0 * n1 * c * c + 4.0 * n1 * n1 * c)  # transformer 2: proj_in + attn1 of its first block


def time_kernel(fn, iters=20):
    """average duration (ms) of one launch, HIP events on the launching stream"""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def time_kernel_graphed(fn, iters=20, reps=3):
    """average duration (ms) of one launch: `iters` launches captured into one hipGraph, replayed `reps` times, HIP events
    on the replaying stream"""
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * reps)


class _Bracket:
    """wraps ``ops.<name>`` so that every call accepted by ``pred`` is bracketed by a pair of HIP events recorded on the launching
    stream; ``external`` events become event-record NODES when the call is captured into a hipGraph, so the pair times that one
    launch inside every replay of the graph"""

    def __init__(self, ops, name, pred, external):
        self.ops, self.name, self.pred, self.external, self.pairs = ops, name, pred, external, []

    def __enter__(self):
        self.real = getattr(self.ops, self.name)

        def wrapped(*a, **kw):
            if not self.pred(*a, **kw):
                return self.real(*a, **kw)
            kw_ev = dict(enable_timing=True, external=True) if self.external else dict(enable_timing=True)
            e0, e1 = torch.cuda.Event(**kw_ev), torch.cuda.Event(**kw_ev)
            e0.record()
            out = self.real(*a, **kw)
            e1.record()
            self.pairs.append((e0, e1))
            return out

        setattr(self.ops, self.name, wrapped)
        return self

    def __exit__(self, *exc):
        setattr(self.ops, self.name, self.real)

    def avg_us(self, overhead_us=0.0):
        return sum(a.elapsed_time(b) for a, b in self.pairs) * 1e3 / max(len(self.pairs), 1) - overhead_us


def event_pair_overhead_us(n=40):
    """what a pair of HIP events recorded back to back on one stream reads with NOTHING between them (median): subtracted from the
    eager per-launch timings (an event record is a packet of its own on the queue)"""
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


def in_step_launch_times(step, ops, probes, reps=3):
    """Average duration of selected launches INSIDE the step, measured live: a replica of the captured step is captured once more
    with event-record nodes around the selected launches (HIP events on the capture stream = the stream the kernels are launched
    on) and replayed ``reps`` times after the timed region (the timed graph itself carries no events).  probes: {label: (ops
    function name, predicate)}.  Returns {label: {"avg_us", "launches"}} and how it was measured; falls back to plain events around
    the same launches of one eager step if event nodes cannot be captured."""
    def measure(external):
        import contextlib
        with contextlib.ExitStack() as st:
            br = {k: st.enter_context(_Bracket(ops, fn, pred, external)) for k, (fn, pred) in probes.items()}
            if external:
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2):
                    step()
                for _ in range(reps):
                    g2.replay()
            else:
                step()
            torch.cuda.synchronize()
            ov = 0.0 if external else event_pair_overhead_us()
            return {k: {"avg_us": round(b.avg_us(ov), 2), "launches": len(b.pairs), "event_pair_overhead_us": round(ov, 2)} for k, b in br.items()}
    try:
        return measure(True), "HIP event-record nodes around each launch inside a replica of the captured step (hipGraph replay)"
    except Exception as e:  # noqa: BLE001 -- event nodes unsupported: eager events (same kernels, same order, host-paced)
        torch.cuda.synchronize()
        return measure(False), ("HIP events around each launch of one eager step of the same kernels in the same order, minus the reading of an "
                                "empty event pair (event-record nodes cannot be captured into a hipGraph on ROCm: %r)" % (e,))


def profile_in_step_avg_us(substr):
    """average in-step duration (us) of the kernel whose name contains `substr`, from the committed rocprofv3 kernel-trace
    summary of `bench.py --step-only` (profiles/r*_bench_kernel_stats*.txt: columns pct calls avg_us min_us max_us name; newest
    first), or (None, None)"""
    import glob
    pdir = os.path.join(ROOT, "profiles")
    for path in sorted(glob.glob(os.path.join(pdir, "r*_bench_kernel_stats*.txt")), reverse=True):
        with open(path) as f:
            for line in f:
                if substr in line and not line.startswith("#"):
                    return float(line.split()[2]), os.path.relpath(path, ROOT)
    return None, None


def profile_top_line():
    """the first entry of the newest committed rocprofv3 kernel-trace summary of `bench.py --step-only`, whatever kernel it is: share of the
    kernel time, launches per step, average duration"""
    import glob
    import re
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_kernel_stats*.txt")), reverse=True):
        steps = None
        with open(path) as f:
            for line in f:
                m = re.search(r"over (\d+) steps", line)
                if m:
                    steps = int(m.group(1))
                if not line.startswith("#") and line.strip():
                    t = line.split(None, 5)
                    return {"kernel": t[5].strip()[:100], "pct_of_kernel_time": float(t[0]), "launches_per_step": round(int(t[1]) / steps, 1) if steps else None,
                            "avg_us": float(t[2]), "live": False, "source": os.path.relpath(path, ROOT)}
    return None


# algorithmic work of the 64-token level's row-tile kernels per launch at B' sample-forwards (N = 64, C = 640; csrc/hsattn.hip)
LEVEL64 = {"hs_attn_kernel<0, true": ("self-attention sub-layer part 1: LN + q|k|v + attention (apad_hs_attention)", lambda b: 2.0 * b * 64 * 640 * 1920 + 4.0 * b * 8 * 64 * 64 * 80),
           "hs_out_kernel": ("to_out + bias + residual (apad_hs_out)", lambda b: 2.0 * b * 64 * 640 * 640),
           "hs_geglu_kernel": ("LN + GEGLU projection (apad_hs_geglu)", lambda b: 2.0 * b * 64 * 640 * 5120),
           "hs_ff2_kernel": ("FF2 + bias + residual (apad_hs_ff2)", lambda b: 2.0 * b * 64 * 2560 * 640)}


def level64_from_profile(B2, n_streams):
    """in-step averages of the 64-token level's kernels from the committed profile -> achieved TFLOP/s (each launch covers B' / streams samples)"""
    out = {}
    for sub, (what, fl) in LEVEL64.items():
        us, src = profile_in_step_avg_us(sub)
        if us:
            b = B2 // max(n_streams, 1)
            out[sub.split("<")[0]] = {"what": what, "in_step_avg_us": us, "samples_per_launch": b, "tflops": round(fl(b) / (us * 1e-6) / 1e12, 1),
                                      "mfma_frac": round(fl(b) / (us * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS, 4), "live": False, "source": src}
    return out


def whole_step_hbm(ms_per_step, B2=64, La=32):
    """HBM bytes of one captured step from the committed PMC passes (tools/round_pmc_traffic.sh) -> fraction of the 8 TB/s peak at this step time"""
    import glob
    for tp in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True):
        with open(tp) as f:
            d = json.load(f)
        ws = d.get("whole_step")
        if ws:
            gbs = ws["bytes_per_step"] / (ms_per_step * 1e-3) / 1e9
            alg = unet_min_bytes_per_step(B2, La)
            return {"bytes_per_step": int(ws["bytes_per_step"]), "read_bytes": int(ws["read_bytes_per_step"]), "write_bytes": int(ws["write_bytes_per_step"]),
                    "achieved_GBs": round(gbs, 1), "hbm_frac_whole_step": round(gbs / HBM_PEAK_GBS, 4),
                    "algorithmic_bytes": int(alg), "measured_over_algorithmic": round(ws["bytes_per_step"] / alg, 3),
                    "algorithmic_bytes_is": "sum over the step's launches of their minimal operand bytes (bench.py::unet_min_bytes_per_step; the CFG-shared "
                                            "prefix counted at the full batch)",
                    "live": False, "source": os.path.relpath(tp, ROOT)}
    return None


def precision_leg(A, unet, inp, args, dev, dtype, steps, graph):
    """the SAME batch-32 step with the model converted (in place) to ``dtype``: ms per step (hipGraph replay when ``graph``, else eager) and the
    guided noise_pred of the FIRST step (fp32 copy) -- the north-star tensor, compared across modes by the caller.  The weights are the
    bf16-rounded synthetic weights in every mode (bf16 -> f16 / f32 is exact for them), so the differences are arithmetic precision only."""
    from ap_adapter_amd import ops
    unet = unet.to(dtype)
    unet.set_kv_cache(False)
    unet.set_kv_cache(True)
    pipe = A.AudioLDM2Pipeline(unet)
    B = args.batch
    ge = pipe.assemble_condition(inp["generated_prompt_embeds"].to(dev), inp["audio_tokens"].to(dev), inp["uncond_audio_tokens"].to(dev), dtype)
    pe, am = inp["prompt_embeds"].to(dev, dtype), inp["attention_mask"].to(dev)
    H, W, Cc = 250, 16, 8
    sched = pipe.scheduler
    sched.set_timesteps(DDIM_STEPS_PER_CLIP)
    coef = sched.coef_table().to(dev)
    step_ptr = torch.zeros(1, dtype=torch.int32, device=dev)
    lat = inp["latents"].to(dev).float().permute(0, 2, 3, 1).reshape(B, H * W, Cc).contiguous()
    lat0 = lat.clone()
    unet_in = lat.to(dtype)
    unet.precompute_time_tables(sched.timesteps.to(dev), step_ptr)

    def step():
        eps2 = unet.forward_nhwc(unet_in, H, W, None, ge, pe, None, am, batch_repeat=2)
        ops.cfg_ddim_step(eps2, lat, unet_in, coef, step_ptr, args.guidance)
        ops.step_advance(step_ptr)

    with torch.no_grad():
        e2 = unet.forward_nhwc(unet_in, H, W, None, ge, pe, None, am, batch_repeat=2).float()
        noise_pred = e2[:B] + args.guidance * (e2[B:] - e2[:B])  # pipeline_audioldm2.py:1021-1022
        if graph:
            s_ = torch.cuda.Stream()
            s_.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s_):
                step()
            torch.cuda.current_stream().wait_stream(s_)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                step()
            run1 = g.replay
        else:
            step()
            run1 = step
        lat.copy_(lat0); unet_in.copy_(lat0.to(dtype)); step_ptr.zero_()
        run1()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run1()
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return {"ms_per_step": round(ms, 2 if graph else 1), "clips_per_s": round(B / (DDIM_STEPS_PER_CLIP * ms * 1e-3), 4), "steps": steps,
            "launch": "hipGraph replay" if graph else "eager", "finite": bool(torch.isfinite(lat).all().item())}, noise_pred


def dominant_kernel_roofline(dev, dtype, B2, in_step=None, in_step_how=None):
    """Roofline of the kernel with the largest share of the step in the committed rocprof summary: the self-attention sub-layer of the
    1000-token level up to to_out -- LayerNorm + to_q | to_k | to_v + softmax(Q K^T / sqrt(32)) V in ONE launch since round 4
    (apad_self_attention_fused, sattn_fused_kernel<bf16, 32>; until round 3 two launches, of which the attention core alone was this
    object).  Algorithmic FLOPs per launch = attention core 4 N^2 C B' + projections 6 N C^2 B'; algorithmic bytes = x read + O written +
    the weights (q, k, v never reach HBM); bound = MFMA (AI = 1370 F/B > ridge 310).
    ONE source for `achieved` / `frac` / `avg_launch_ms`: the kernel's average duration INSIDE the step, measured live by
    in_step_launch_times (HIP events on the launching stream).  Beside it: `frac_isolated` (20 back-to-back launches inside one hipGraph)
    and `rocprof_in_step_avg_us`, the same kernel's average in the committed `rocprofv3 --kernel-trace --stats` summary of
    `bench.py --step-only`, which must agree with `avg_launch_ms`."""
    from ap_adapter_amd import ops
    N, C, heads = 1000, 256, 8
    x = torch.randn(B2, N, C, device=dev).to(dtype)
    ln = (torch.ones(C, device=dev, dtype=dtype), torch.zeros(C, device=dev, dtype=dtype), 1e-5)
    wq, wk, wv = ((torch.randn(C, C, device=dev) * 0.02).to(dtype) for _ in range(3))
    pk, csbb = ops.sattn_pack(wq, wk, wv, ln, heads)
    out = torch.empty_like(x)
    ms = time_kernel_graphed(lambda: ops.self_attention_fused(x, pk, csbb, heads, 1e-5, out=out))
    flops_core, flops_proj = 4.0 * N * N * C * B2, 6.0 * N * C * C * B2
    flops = flops_core + flops_proj
    nbytes = 2 * B2 * N * C * 2 + 3 * C * C * 2
    ach = flops / (ms * 1e-3) / 1e12
    # HBM bytes per launch from the PMC passes (tools/round_pmc_traffic.sh: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 runs over the
    # step, counter units calibrated by a known-size probe in the same run); the counters cannot be collected from inside this process,
    # so the committed measurement is reported when it is for this exact launch
    traffic = tsrc = None
    import glob
    for tp in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True):
        if dtype == torch.bfloat16 and B2 == 64:
            with open(tp) as f:
                for k in json.load(f).get("per_kernel", []):
                    if SELF_ATTN_KERNEL_SUBSTR in k["kernel"]:
                        traffic, tsrc = int(k["read_bytes_per_launch"] + k["write_bytes_per_launch"]), os.path.relpath(tp, ROOT)
            if traffic:
                break
    prof_us, psrc = profile_in_step_avg_us(SELF_ATTN_KERNEL_SUBSTR)
    if in_step is not None and in_step["launches"] > 0:
        ms_src, how, n_l = in_step["avg_us"] * 1e-3, in_step_how, in_step["launches"]
    else:  # (--no-in-step: only the isolated timing exists)
        ms_src, how, n_l = ms, "isolated re-timing (20 launches in one hipGraph, HIP events)", 20
    ach_src = flops / (ms_src * 1e-3) / 1e12
    return {"kernel": "LayerNorm + q|k|v + self-attention of the 1000-token level in one launch (apad_self_attention_fused, d=32) B'=%d heads=8 N=L=1000" % B2,
            "bound": "mfma", "achieved": round(ach_src, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach_src / MFMA_PEAK_TFLOPS, 4), "avg_launch_ms": round(ms_src, 4), "avg_launch_is": how, "launches_timed": n_l,
            "frac_isolated": round(ach / MFMA_PEAK_TFLOPS, 4), "isolated_avg_launch_ms": round(ms, 4),
            "rocprof_in_step_avg_us": prof_us, "rocprof_source": psrc,
            "live": {"avg_launch_ms": True, "frac": True, "isolated_avg_launch_ms": True, "rocprof_in_step_avg_us": False, "traffic": False},
            "flops_per_launch": flops, "flops_attention_core": flops_core, "flops_projections": flops_proj, "algorithmic_bytes": nbytes, "traffic": traffic,
            "traffic_source": (tsrc + " (rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes over the step)") if traffic else None}


def mfma_ceiling(dev):
    """What the matrix pipe of THIS device delivers on a memory-free stream of dense bf16 MFMAs (apad_probe_mfma: four chains per wave, two waves per
    SIMD), on zero operands and on pseudo-random ones, ~15 ms launches timed with HIP events: the rate is power-managed and depends on the operand data
    (tools/ubench/mfma_data.hip).  `peak` in the roofline objects stays the nominal 2.5 PF; this object says how much of it real activations can reach."""
    import ctypes as C
    from ap_adapter_amd import _lib as L
    h = L.lib()
    sink = torch.zeros(4, device=dev)
    out = {}
    for mode, name in ((0, "zero_operands"), (1, "random_operands")):
        fl = C.c_double(0.0)
        st = torch.cuda.current_stream().cuda_stream
        L.check(h.apad_probe_mfma(sink.data_ptr(), mode, 200, C.byref(fl), st), "apad_probe_mfma")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(h.apad_probe_mfma(sink.data_ptr(), mode, 12000, C.byref(fl), st), "apad_probe_mfma")
        e1.record()
        torch.cuda.synchronize()
        tf = fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12
        out[name] = {"tflops": round(tf, 1), "frac_of_nominal": round(tf / MFMA_PEAK_TFLOPS, 4)}
    out["what"] = ("memory-free dense bf16 MFMA stream (v_mfma_f32_32x32x16_bf16, 4 chains per wave, 2 waves per SIMD), measured live on this device: the "
                   "rate on random operands is the ceiling of every MFMA-bound kernel of the step (power-managed clocks)")
    return out


def fused_attn2_roofline(dev, dtype, B2, La, ap_scale, in_step=None, in_step_how=None):
    """The adapter's own kernel, the one the north star names: apad_fused_cross_attention -- LayerNorm + to_q + decoupled
    attention (8 text + La audio keys, two softmaxes blended by ap_scale) + to_out + bias + residual of one attn2
    sub-layer in ONE launch, at the 1000-token level (the 10 adapted + 10 T5 sites that cost most).
    Algorithmic FLOPs: q + out projections 4 N C^2 + attention core 4 N (8 + La) C per sample;
    algorithmic bytes: read x, write out (2 x N C 2 B per sample) + both weights + the hoisted K / V.  AI = 285 F/B at
    La = 32, just under the 310 F/B ridge: both fractions are reported, the larger one names the binding roof."""
    from ap_adapter_amd import ops
    N, C, heads, Lt = 1000, 256, 8, 8
    R = lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(dtype)
    x, g_, b_ = R(B2, N, C), R(C), R(C)
    wq, wo, bo = R(C, C, std=0.02), R(C, C, std=0.02), R(C, std=0.02)
    k1, k2 = R(B2, Lt, C, std=0.3), R(B2, La, C, std=0.3)
    v1t = torch.zeros(B2, heads, C // heads, 32, device=dev, dtype=dtype)
    v2t = torch.zeros(B2, heads, C // heads, ops.round_up(La, 32), device=dev, dtype=dtype)
    v1t[..., :Lt].normal_(0, 0.3)
    v2t[..., :La].normal_(0, 0.3)
    out = torch.empty_like(x)
    flops = (4.0 * N * C * C + 4.0 * N * (Lt + La) * C) * B2
    nbytes = 2 * B2 * N * C * 2 + 2 * C * C * 2 + 2 * B2 * (Lt + La) * C * 2
    if ops.xattn_lengths_ok(Lt, La):
        (wq_p, q_fold), wo_p = ops.xattn_pack_weight(wq, (g_, b_, 1e-5)), ops.xattn_pack_weight(wo)
        pk1, pk2 = ops.xattn_pack_kv(k1, v1t, Lt), ops.xattn_pack_kv(k2, v2t, La)
        fn = lambda: ops.fused_cross_attention(x, wq_p, wo_p, bo, pk1, Lt, heads, ln=(g_, b_, 1e-5), kv2_packed=pk2, L2=La,
                                               scale2=ap_scale, out=out, q_fold=q_fold)
        name = "xattn_kernel<bf16> (apad_fused_cross_attention): LN + to_q + decoupled attention + to_out + residual"
    else:  # outside the one-launch kernel's key counts (La = 512): the three-kernel chain (LN + to_q ; decoupled attention ; to_out + residual)
        def fn():
            qd = ops.fused_linear(x, wq, ln=(g_, b_, 1e-5))
            od = ops.attention(qd, k1, v1t, Lt, heads, k2=k2, vt2=v2t, L2=La, scale2=ap_scale)
            ops.fused_linear(od, wo, bo, residual=x, out=out)
        name = "three launches (rowpanel LN+to_q ; attn_kernel<DUAL> ; rowpanel to_out+residual)"
        nbytes += 4 * B2 * N * C * 2
    ms = time_kernel_graphed(fn)
    tf, gbs = flops / (ms * 1e-3) / 1e12, nbytes / (ms * 1e-3) / 1e9
    iso = {"avg_launch_ms": round(ms, 4), "mfma_frac": round(tf / MFMA_PEAK_TFLOPS, 4), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4)}
    how = "isolated re-timing (20 launches in one hipGraph, HIP events)"
    if in_step is not None and in_step["launches"] > 0:  # ONE source for the fractions: the live in-step average
        ms, how = in_step["avg_us"] * 1e-3, in_step_how
        tf, gbs = flops / (ms * 1e-3) / 1e12, nbytes / (ms * 1e-3) / 1e9
    return {"kernel": name + " B'=%d N=1000 C=256 Lt=8 La=%d" % (B2, La),
            "bound": "mfma" if tf / MFMA_PEAK_TFLOPS >= gbs / HBM_PEAK_GBS else "hbm",
            "mfma": {"achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS, 4)},
            "hbm": {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)},
            "avg_launch_ms": round(ms, 4), "avg_launch_is": how, "launches_timed": (in_step or {}).get("launches", 20),
            "isolated": iso, "flops": flops, "algorithmic_bytes": nbytes}


def fused_attn2_grid(dev, dtype, B2, ap_scale):
    """the adapter's attn2 sub-layer (LayerNorm + to_q + text/audio softmaxes blended by ap_scale + to_out + bias +
    residual; K/V hoisted) at every adapted level x every pooling setting of cfg 3's sweep, entered exactly as the UNet enters it
    (Attention module + IPAttnProcessor2_0 with residual / ln, hoist on), so each cell times whatever route the processor takes
    there.  Isolated timing (20 calls in one hipGraph).  FLOPs 4 N C^2 + 4 N (8 + La) C, bytes 2 x N C 2 + weights + K/V per sample."""
    import ap_adapter_amd as A
    from ap_adapter_amd import ops
    from ap_adapter_amd.unet import Attention
    from ap_adapter_amd.synthetic import init_synthetic_
    grid = {}
    for C_, N in ((256, 1000), (384, 252), (640, 64)):
        with torch.device(dev):
            attn = Attention(C_, 768, 8, C_ // 8)
            proc = A.IPAttnProcessor2_0(hidden_size=C_, name="bench", cross_attention_dim=768, num_tokens=8, scale=ap_scale)
        attn.set_processor(proc)
        init_synthetic_(attn, 5, on_device=True)
        attn = attn.to(dev, dtype).requires_grad_(False)
        proc.kv_cache_enabled = True
        x = torch.randn(B2, N, C_, device=dev).to(dtype)
        ln = (torch.ones(C_, device=dev, dtype=dtype), torch.zeros(C_, device=dev, dtype=dtype), 1e-5)
        for La in (8, 32, 128, 256, 512):
            ehs = torch.randn(B2, 8 + La, 768, device=dev).to(dtype)
            calls, hs = [], []
            real, real_rows, real_hs = ops.fused_cross_attention, ops.cross_attention_rows, ops.hs_attention
            ops.fused_cross_attention = lambda *a, **kw: (calls.append(1), real(*a, **kw))[1]
            ops.cross_attention_rows = lambda *a, **kw: (calls.append(1), real_rows(*a, **kw))[1]  # (the 384-wide level's one-launch kernel)
            ops.hs_attention = lambda *a, **kw: (hs.append(1), real_hs(*a, **kw))[1]               # (the 64-token level: head-sliced, + hs_out)
            try:
                with torch.no_grad():
                    ms = time_kernel_graphed(lambda: attn(x, encoder_hidden_states=ehs, residual=x, ln=ln))
            finally:
                ops.fused_cross_attention, ops.cross_attention_rows, ops.hs_attention = real, real_rows, real_hs
            flops = (4.0 * N * C_ * C_ + 4.0 * N * (8 + La) * C_) * B2
            nbytes = 2 * B2 * N * C_ * 2 + 2 * C_ * C_ * 2 + 2 * B2 * (8 + La) * C_ * 2
            grid[f"C{C_}_N{N}_La{La}"] = {"us": round(ms * 1e3, 2), "route": "one launch" if calls else ("two launches (head-sliced + to_out)" if hs else "three launches"),
                                         "mfma_frac": round(flops / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                                         "hbm_frac": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            proc.clear_kv_cache()
    return grid


def audiomae_ms(dev, dtype=torch.float32):
    """AudioMAE ViT-B (12 blocks, 513 tokens) + (avg + max)/2 pooling over 2 mels [1024, 128], random-init weights, in the
    reference's own arithmetic type (fp32, pipeline_audioldm2.py:926: exact-f32 MFMA mode)"""
    import ap_adapter_amd as A
    from ap_adapter_amd.synthetic import init_synthetic_
    with torch.device(dev):
        m = A.AudioMAEConditionCTPoolRand()
    init_synthetic_(m, 7, w_std=0.02, on_device=True)
    m = m.to(dev, dtype)
    mel = torch.randn(2, 1024, 128, device=dev) * 0.5
    with torch.no_grad():
        return time_kernel(lambda: m(mel, time_pool=4, freq_pool=4), iters=5)


def cpu_baseline():
    """SURVEY 8d "CPU baseline beside it": the reference-equivalent CPU path (the oracle restatement, fp32 torch; kind
    "port") on this box's host cores, BASELINE cfg 1 EXACTLY -- timbre_transfer preset (ap_scale 0.5, pooling 2x2 -> La = 128,
    guidance 7.5), one 10 s clip, 5 DDIM steps with CFG (10 UNet sample-forwards) -- extrapolated x40 to the 200-step
    metric; the same step at ONE thread (1 DDIM step, x200); and the per-layer processor micro-benchmark of SURVEY 6."""
    import ap_adapter_amd as A
    from ap_adapter_amd.synthetic import init_synthetic_, synthetic_inputs
    from oracle import unet as OU, ddim
    from oracle.attention import ip_attn_processor_2_0
    La, gs, scale, steps = 128, 7.5, 0.5, 5
    u = A.AudioLDM2UNet2DConditionModel()
    A.install_ap_adapter(u, None, scale=scale)
    init_synthetic_(u, 100)
    sd = {k: v.detach() for k, v in u.state_dict().items()}
    procs = {n: dict(scale=p.scale, num_tokens=p.num_tokens) for n, p in u.attn_processors.items() if hasattr(p, "to_k_ip")}
    cfg = u.config.geometry_dict()
    inp = synthetic_inputs(1, La)
    ehs = torch.cat([torch.cat([inp["generated_prompt_embeds"][:1], inp["uncond_audio_tokens"]], 1),
                     torch.cat([inp["generated_prompt_embeds"][1:], inp["audio_tokens"]], 1)], 0)
    fn = lambda x, t: OU.unet_forward(sd, cfg, x, t, ehs, inp["prompt_embeds"], None, inp["attention_mask"].float(), procs)
    # a bounded, stable sample: the intra-op pool that is FASTEST on this host among 8 / 16 / 32 threads (one DDIM step each; the 128-thread default
    # oversubscribes the shared host, and on the round-6 hosts 8 threads beat 32 on this operator mix)
    with torch.no_grad():
        best = None
        for cand in sorted({min(torch.get_num_threads(), c) for c in (8, 16, 32)}):
            torch.set_num_threads(cand)
            t0 = time.time()
            ddim.denoise_loop(fn, inp["latents"], 1, gs)
            dt_ = time.time() - t0
            if best is None or dt_ < best[0]:
                best = (dt_, cand)
    cores = best[1]
    torch.set_num_threads(cores)
    with torch.no_grad():
        t0 = time.time()
        ddim.denoise_loop(fn, inp["latents"], steps, gs)
        s_all = (time.time() - t0) / steps
        torch.set_num_threads(1)
        t0 = time.time()
        ddim.denoise_loop(fn, inp["latents"], 1, gs)
        s_one = time.time() - t0
        # SURVEY 6 micro-benchmark: the decoupled processor alone, B' = 2, at the three adapted levels
        micro = {}
        for threads in (cores, 1):
            torch.set_num_threads(threads)
            for (C_, N) in ((256, 1000), (384, 252), (640, 64)):
                g = torch.Generator().manual_seed(0)
                hs, e = torch.randn(2, N, C_, generator=g), torch.randn(2, 8 + La, 768, generator=g)
                W = lambda o, i: torch.randn(o, i, generator=g) * 0.02
                args = (hs, e, W(C_, C_), W(C_, 768), W(C_, 768), W(C_, C_), torch.zeros(C_), W(C_, 768), W(C_, 768), 8, 8, scale)
                for _ in range(3):
                    ip_attn_processor_2_0(*args)
                t0 = time.time()
                for _ in range(10):
                    ip_attn_processor_2_0(*args)
                micro[f"C{C_}_N{N}_La{La}_{threads}thr_ms"] = round((time.time() - t0) / 10 * 1e3, 3)
        torch.set_num_threads(cores)
    return {"value": round(1.0 / (DDIM_STEPS_PER_CLIP * s_all), 6), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"reference-equivalent CPU path (oracle restatement, fp32 torch), BASELINE cfg 1 exactly: timbre_transfer preset "
                      f"(ap_scale {scale}, La={La}, guidance {gs}), batch 1, {steps} DDIM steps with CFG = {2 * steps} UNet "
                      f"sample-forwards ({s_all:.2f} s/step on {cores} threads), extrapolated x{DDIM_STEPS_PER_CLIP // steps} to 200 steps",
            "one_thread": {"value": round(1.0 / (DDIM_STEPS_PER_CLIP * s_one), 6), "unit": "clips/s", "cores": 1,
                           "sample": f"same configuration, 1 DDIM step ({s_one:.2f} s) x{DDIM_STEPS_PER_CLIP}"},
            "processor_microbench_ms": micro}


def train_measure(args, A, rank, world, dev, steps, warmup):
    """BASELINE cfg 5 -- the adapter's training step (train_apadapter_v2.py:892-979) at per-GPU batch 4: UNet forward
    at a random t per sample, fp32 MSE, backward to the 64 adapter tensors, ONE flat fp32 all-reduce (RCCL) when N > 1, clip +
    AdamW; the micro-step replays as one hipGraph.  One bench step = one optimizer step; value = samples/s over all ranks."""
    import torch.distributed as dist
    from ap_adapter_amd.synthetic import init_synthetic_
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.train_dtype]  # f32 = the reference's default (train.sh)
    B, La = args.train_batch, args.la
    with torch.device(dev):
        u = A.AudioLDM2UNet2DConditionModel()
        A.install_ap_adapter(u, None, scale=0.5)
    init_synthetic_(u, 100, on_device=True)
    u = u.to(dev, dtype)
    tr = A.AdapterTrainer(u, lr=1e-4)
    g = torch.Generator().manual_seed(1000 * rank)
    lat = torch.randn(B, 8, 250, 16, generator=g).to(dev)
    ehs = torch.randn(B, 8 + La, 768, generator=g).to(dev)
    ehs1 = torch.randn(B, 16, 1024, generator=g).to(dev)
    m1 = torch.ones(B, 16, device=dev)
    replay = tr.capture_micro_step(B, 250, 16, 8 + La, 16)
    losses = []

    def one():
        noise = torch.randn(B, 8, 250, 16, generator=g).to(dev)
        t = torch.randint(0, 1000, (B,), generator=g).to(dev)
        losses.append(replay(A.add_noise(lat, noise, t, tr.alphas_cumprod), t, ehs, ehs1, m1, noise).clone())
        tr.optimizer_step()  # (all-reduce of the flat gradient inside, when world > 1)

    for _ in range(warmup):
        one()
    dt, every = A.distributed.timed_steps(lambda n: [one() for _ in range(n)], steps, world, torch.cuda.synchronize, dev)
    ms = dt / steps * 1e3
    accum = None
    k = int(getattr(args, "train_accum", 0) or 0)
    if k > 1:
        # train.sh's gradient_accumulation_steps=4 two ways (same optimizer step, tests/test_gpu_train.py): the reference's k micro-batches
        # of B one after another, and ONE pass over their concatenation (AdapterTrainer.micro_step(micro_batches=k))
        def window(rp, nb, reps):
            noise = torch.randn(nb, 8, 250, 16, generator=g).to(dev)
            t = torch.randint(0, 1000, (nb,), generator=g).to(dev)
            r = lambda x: x.repeat(nb // B, *([1] * (x.dim() - 1)))
            for _ in range(reps):
                rp(A.add_noise(r(lat), noise, t, tr.alphas_cumprod), t, r(ehs), r(ehs1), r(m1), noise)
            tr.optimizer_step()
        replay_k = tr.capture_micro_step(B * k, 250, 16, 8 + La, 16, micro_batches=k)
        res = {}
        for name, rp, nb, reps in (("sequential", replay, B, k), ("as_batch", replay_k, B * k, 1)):
            window(rp, nb, reps)
            dtk, _ = A.distributed.timed_steps(lambda n: [window(rp, nb, reps) for _ in range(n)], max(steps // 2, 2), world, torch.cuda.synchronize, dev)
            msk = dtk / max(steps // 2, 2) * 1e3
            res[name] = {"ms_per_optimizer_step": round(msk, 3), "samples_per_s": round(B * k * world / (msk * 1e-3), 3)}
        accum = {"gradient_accumulation_steps": k, "per_gpu_micro_batch": B, **res,
                 "note": "train.sh runs accumulation 4; one pass over the 4 concatenated micro-batches is the same optimizer step"}
    return {"metric": "adapter training samples/sec, AudioLDM2-large+AP (BASELINE cfg 5)", "value": round(B * world / (ms * 1e-3), 3),
            "accumulation": accum,
            "unit": "samples/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 3),
            "rank_ms_per_step": [round(t / steps * 1e3, 3) for t in every],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.train_dtype, "data": "synthetic",
            "config": {"workload": f"train_apadapter_v2.py step: batch {B}/GPU, random t per sample, La={La}, {args.train_dtype} compute, fp32 master + "
                                   f"AdamW, adapter-only gradients (21 626 880 parameters), micro-step replayed as one hipGraph",
                       "global_batch": B * world, "parallelism": f"dp{world} (one flat 86.5 MB fp32 all-reduce per optimizer step)"},
            "loss_first": float(losses[0]), "loss_last": float(losses[-1]), "finite": bool(torch.isfinite(torch.stack(losses)).all())}


def train_main(args, A, rank, world, dev):
    """--train: the training step as the bench's step (same launch contract); prints ONE JSON line on rank 0"""
    import torch.distributed as dist
    line = train_measure(args, A, rank, world, dev, args.steps, args.warmup)
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def stub_main(args):
    """--stub-step-ms: the launch / timing contract alone, no GPU and no model -- every rank sleeps (1 + rank / 2) x the given
    milliseconds per step over gloo.  What tests/test_distributed.py drives to check that `bench.py --gpus N` starts N ranks by itself,
    that they all join, and that the reported time is the MAX over ranks."""
    import ap_adapter_amd as A
    import torch.distributed as dist
    rank, world, _ = A.distributed.init_from_env("gloo")
    per = args.stub_step_ms * (1.0 + 0.5 * rank) * 1e-3
    dt, every = A.distributed.timed_steps(lambda n: [time.sleep(per) for _ in range(n)], args.steps, world, lambda: None)
    if rank == 0:
        ms = dt / args.steps * 1e3
        print(json.dumps({"metric": "stub", "stub": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(ms, 3), "rank_ms_per_step": [round(t / args.steps * 1e3, 3) for t in every],
                          "value": round(args.batch * world / (DDIM_STEPS_PER_CLIP * ms * 1e-3), 4), "unit": "clips/s", "scaling": "weak"}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--la", type=int, default=32, help="audio tokens (32 = style_transfer preset, pooling 4x4)")
    ap.add_argument("--guidance", type=float, default=9.5)
    ap.add_argument("--ap-scale", type=float, default=0.55)
    ap.add_argument("--low-res-streams", type=int, default=1, help="n > 0: run the two batch halves of the n lowest-resolution levels on two streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the fp32-precision-mode timing of the same step")
    ap.add_argument("--step-only", action="store_true", help="only the timed step (for rocprofv3 runs: no roofline re-timings, no AudioMAE, no CPU leg)")
    ap.add_argument("--train", action="store_true", help="time BASELINE cfg 5 (the adapter's training step) instead of the denoise step")
    ap.add_argument("--train-batch", type=int, default=4)
    ap.add_argument("--train-dtype", choices=["bf16", "f16", "f32"], default="bf16", help="compute type of the training step (fp32 master weights in every mode)")
    ap.add_argument("--train-accum", type=int, default=4, help="also time an accumulation window of this many micro-batches, sequential and as one batch (0 = skip)")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the `train` sub-object (cfg 5: 5 optimizer steps at batch 4) of the default line")
    ap.add_argument("--no-in-step", action="store_true", help="skip the live in-step kernel timing (roofline falls back to the isolated timing)")
    ap.add_argument("--stub-step-ms", type=float, default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()

    from ap_adapter_amd import distributed as D
    if args.gpus > 1 and not D.launched_by_torchrun():
        # `python bench.py --gpus N` typed without a launcher: start the N ranks of this node ourselves (one per GPU) with the
        # driver's own command; rank 0 of that job prints the line
        sys.exit(D.spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    if args.stub_step_ms is not None:
        return stub_main(args)

    import ap_adapter_amd as A
    from ap_adapter_amd import ops
    from ap_adapter_amd.synthetic import init_synthetic_, synthetic_inputs
    import torch.distributed as dist

    rank, world, local = A.distributed.init_from_env("nccl" if args.gpus > 1 else None)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.train:
        return train_main(args, A, rank, world, dev)
    dtype = torch.bfloat16
    B = args.batch

    t_build = time.perf_counter()
    with torch.device(dev):  # parameters are created and initialised ON the device (718 M parameters: a second, not a minute)
        unet = A.AudioLDM2UNet2DConditionModel()
        A.install_ap_adapter(unet, None, scale=args.ap_scale)
    init_synthetic_(unet, 100, on_device=True)
    unet = unet.to(dev, dtype)
    torch.cuda.synchronize()
    build_ms = (time.perf_counter() - t_build) * 1e3
    inp = synthetic_inputs(B, args.la, seed=1000 * rank)
    pipe = A.AudioLDM2Pipeline(unet)
    ge = pipe.assemble_condition(inp["generated_prompt_embeds"].to(dev), inp["audio_tokens"].to(dev),
                                 inp["uncond_audio_tokens"].to(dev), dtype)
    pe = inp["prompt_embeds"].to(dev, dtype)
    am = inp["attention_mask"].to(dev)
    H, W, Cc = 250, 16, 8
    sched = pipe.scheduler
    sched.set_timesteps(DDIM_STEPS_PER_CLIP)
    coef = sched.coef_table().to(dev)
    step_ptr = torch.zeros(1, dtype=torch.int32, device=dev)
    lat = inp["latents"].to(dev).float().permute(0, 2, 3, 1).reshape(B, H * W, Cc).contiguous()
    lat0 = lat.clone()
    unet_in = lat.to(dtype)
    # ---- one-off setup of a pipeline call, outside the timed region (reported as setup_ms): time tables of the 22 resnets for
    #      all 200 steps, K/V hoist + packing of the 64 cross-attention sites (first step), warm-up step, hipGraph capture ----
    torch.cuda.synchronize()
    t_setup = time.perf_counter()
    unet.set_kv_cache(True)
    unet.precompute_time_tables(sched.timesteps.to(dev), step_ptr)

    if args.low_res_streams:
        # (round 5 default: ONE stream -- the 64-token level's launches are 4 x 64 = 256 workgroups at the full CFG batch and fill the chip by
        #  themselves; same-box A/B 37.49 / 37.54 ms with two half-batch streams vs 37.35 / 37.36 with one, and half the launches at that level)
        unet.low_res_streams = tuple(torch.cuda.Stream() for _ in range(int(os.environ.get("APAD_LOW_RES_NSTREAMS", "1"))))
        unet.low_res_levels = args.low_res_streams

    def step():
        # (round 2 also ran the two CFG halves as two concurrent batch-32 forwards on forked streams: +1.5 % step time, removed)
        eps2 = unet.forward_nhwc(unet_in, H, W, None, ge, pe, None, am, batch_repeat=2)
        ops.cfg_ddim_step(eps2, lat, unet_in, coef, step_ptr, args.guidance)
        ops.step_advance(step_ptr)

    with torch.no_grad():
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
    torch.cuda.synchronize()
    setup_ms = (time.perf_counter() - t_setup) * 1e3

    def reset():
        lat.copy_(lat0)
        unet_in.copy_(lat0.to(dtype))
        step_ptr.zero_()

    def run(n):
        for i in range(n):
            if int(i) % DDIM_STEPS_PER_CLIP == 0:
                reset()  # a new batch of clips starts every 200 steps (table index stays in range)
            g.replay()

    reset()
    run(args.warmup)
    reset()
    dt, every = D.timed_steps(run, args.steps, world, torch.cuda.synchronize, dev)  # barrier + sync | K steps | sync + barrier; MAX over ranks
    finite = bool(torch.isfinite(lat).all().item())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        clips_per_s = (B * world) / (DDIM_STEPS_PER_CLIP * ms_per_step * 1e-3)
        from ap_adapter_amd import unet as _U
        shared = bool(_U.CFG_SHARED_PREFIX)
        fl_model = unet_flops_per_sample(args.la) * 2 * B
        fl = fl_model - (cfg_shared_prefix_flops() * B if shared else 0.0)  # EXECUTED FLOPs: the condition-free prefix runs once per clip
        line = {
            "metric": "10s-clips/sec @200 DDIM steps, AudioLDM2-large+AP", "value": round(clips_per_s, 4), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "rank_ms_per_step": [round(t / args.steps * 1e3, 3) for t in every],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_NAME[dtype], "data": "synthetic",
            "config": {"workload": f"AudioLDM2-large geometry + 32 AP processors, style_transfer preset "
                                   f"(ap_scale {args.ap_scale}, La={args.la}, CFG {args.guidance}), batch {B}/GPU, 10 s clips "
                                   f"(latents 8x250x16), 200-step DDIM, hipGraph-captured step; one bench step = one DDIM step",
                       "global_batch": B * world, "parallelism": f"dp{world} (independent clips, no collective in the loop)",
                       "weights": "random-init N(0,0.02^2), real shapes (no checkpoint offline)"},
            "step_tflops": round(fl / (ms_per_step * 1e-3) / 1e12, 2),
            "mfma_frac_whole_step": round(fl / (ms_per_step * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "finite": finite,
            # the CFG batch's two halves are the same rows until the first conditioned attention: that prefix is executed once per
            # clip (bit-identical latents, tests/test_gpu_unet.py::test_cfg_shared_prefix_equals_the_duplicated_batch); step_tflops
            # counts executed FLOPs only
            "cfg_shared_prefix": {"enabled": shared, "flops_not_repeated_per_step": cfg_shared_prefix_flops() * B if shared else 0.0,
                                  "flops_model_per_step": fl_model},
            # one-off per pipeline call, outside the timed region: time tables, K/V hoist + packing, warm-up step, graph capture
            "setup_ms": round(setup_ms, 1), "model_build_ms": round(build_ms, 1),
        }
        if args.step_only:
            print(json.dumps(line))
            return
        ins, how = {}, None
        if not args.no_in_step:
            with torch.no_grad():
                ins, how = in_step_launch_times(step, ops, {
                    # (only the launches at the full CFG batch: the shared prefix's transformer runs its self-attention at half the batch, and the FLOPs per
                    #  launch below are those of 2B sample-forwards)
                    "self_attn_1000": ("self_attention_fused", lambda x, *a, **kw: x.shape[1] == 1000 and x.shape[0] == 2 * B),
                    "fused_attn2_ip": ("fused_cross_attention", lambda x, *a, **kw: kw.get("L2", 0) > 0),
                    "fused_attn2_t5": ("fused_cross_attention", lambda x, *a, **kw: not kw.get("L2", 0))})
        line["roofline"] = dominant_kernel_roofline(dev, dtype, 2 * B, ins.get("self_attn_1000"), how)
        try:
            line["mfma_ceiling"] = mfma_ceiling(dev)
        except Exception as e:  # noqa: BLE001
            line["mfma_ceiling"] = {"error": repr(e)}
        line["roofline_top_line"] = profile_top_line()
        nst = len(unet.low_res_streams) if getattr(unet, "low_res_streams", None) else 1
        line["level64_kernels"] = level64_from_profile(2 * B, nst)
        line["hbm_whole_step"] = whole_step_hbm(ms_per_step, 2 * B, args.la)
        line["fused_attn2"] = fused_attn2_roofline(dev, dtype, 2 * B, args.la, args.ap_scale, ins.get("fused_attn2_ip"), how)
        line["fused_attn2"]["t5_sites_in_step"] = ins.get("fused_attn2_t5")
        try:
            line["fused_attn2"]["grid"] = fused_attn2_grid(dev, dtype, 2 * B, args.ap_scale)
        except Exception as e:  # noqa: BLE001
            line["fused_attn2"]["grid"] = {"error": repr(e)}
        # the audio-condition encoder (2 mels per call: clip + zeros) is outside the timed loop; reported
        # separately and folded into a per-clip "included" figure (one pipeline call of B clips pays it once)
        try:
            mae_ms = audiomae_ms(dev)
            per_call_s = DDIM_STEPS_PER_CLIP * ms_per_step * 1e-3 + mae_ms * 1e-3
            line["audiomae"] = {"ms_per_call": round(mae_ms, 3), "mels": 2, "dtype": "f32 (exact-f32 MFMA; the reference's AudioMAE type)",
                                "clips_per_s_including_it": round(B * world / per_call_s, 4)}
        except Exception as e:  # never lose the headline line over the side measurement
            line["audiomae"] = {"error": repr(e)}
        if not args.no_train_leg and world == 1:
            # BASELINE cfg 5 beside the headline (driver-observed): 5 graph-replayed optimizer steps at per-GPU batch 4
            try:
                targs = argparse.Namespace(**{**vars(args), "la": 32})
                t_ = train_measure(targs, A, rank, world, dev, steps=5, warmup=2)
                line["train"] = {k: t_[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config", "loss_first",
                                                     "loss_last", "finite", "accumulation")}
            except Exception as e:  # noqa: BLE001
                line["train"] = {"error": repr(e)}
        if not args.no_fp32_leg and world == 1:
            # the other two precision modes of the SAME step, beside the headline (all live): f16 -- what the reference ships (inference.py:13,
            # torch_dtype=float16) --, graph-replayed like the bf16 line, and fp32 (exact-f32 MFMA, eager).  noise_pred (the guided epsilon of
            # the first step, batch 32) of the 16-bit modes is compared with the fp32 mode's, which tests/test_gpu_unet.py holds within 1e-3
            # (measured 1e-5) of the oracle chain at the same geometry
            try:
                with torch.no_grad():
                    step_ptr.zero_()  # (the time tables are indexed by the device step counter: the first step, like the other modes below)
                    e2 = unet.forward_nhwc(lat0.to(dtype), H, W, None, ge, pe, None, am, batch_repeat=2).float()
                    np_bf16 = e2[:B] + args.guidance * (e2[B:] - e2[:B])
                f16, np_f16 = precision_leg(A, unet, inp, args, dev, torch.float16, steps=10, graph=True)
                f32, np_f32 = precision_leg(A, unet, inp, args, dev, torch.float32, steps=2, graph=False)
                mx = float(np_f32.abs().max())
                f16["noise_pred_max_abs_vs_fp32_mode"] = round(float((np_f16 - np_f32).abs().max()), 6)
                f16["what"] = "the same batch-32 step in f16 storage (the reference's inference dtype, inference.py:13), hipGraph replay"
                f32["what"] = ("the same batch-32 step with fp32 storage and exact-f32 MFMA (v_mfma_f32_32x32x2_f32, 1/16 of the bf16 rate), eager launches: "
                               "the mode in which noise_pred is within 1e-3 of the oracle chain (measured 1e-5)")
                line["f16_mode"], line["fp32_mode"] = f16, f32
                line["noise_pred_vs_fp32_mode"] = {"bf16_max_abs": round(float((np_bf16 - np_f32).abs().max()), 6), "f16_max_abs": f16["noise_pred_max_abs_vs_fp32_mode"],
                                                   "max_abs_noise_pred": round(mx, 4), "tensor": f"guided noise_pred of the first DDIM step, batch {B}, CFG {args.guidance}"}
            except Exception as e:  # noqa: BLE001
                line["fp32_mode"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:  # the host baseline is reported at N = 1 only
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
