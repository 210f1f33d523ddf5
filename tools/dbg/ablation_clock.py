"""Where does the e4m3 loop's power go?  The dense launch of every -DVORTA_DIAG_* build (wrong results: each removes one
ingredient of the step) on RANDOM operands, with the shader clock and socket power sampled from rocm-smi during the launches.
In the regime the loop runs in (profiles/r03_fp8_loop_experiments.txt) the time of a variant = its cycles / the clock the
chip grants it, so the clock column says what each ingredient costs in power, the cycles column what it costs in schedule.
    for v in NOLDSRD NODMA "NOEXP -DVORTA_DIAG_NOCVT" NOMAX NOMFMA; do n=${v%% *}; VORTA_BUILD_SUFFIX=_d$n \\
        VORTA_EXTRA_FLAGS="-DVORTA_FP8_DIAG -DVORTA_DIAG_$v" python -m vorta_amd.build; done
    python tools/dbg/ablation_clock.py          # runs itself once per library (child processes)"""
import glob, json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        card = next(iter(json.loads(out).values()))
        s = next(v for k, v in card.items() if k.startswith("sclk clock speed"))
        return float(s.strip("()").lower().replace("mhz", "")), next(float(v) for k, v in card.items() if "Power" in k)
    except Exception:
        return None


def one(tag):
    import torch
    from vorta_amd import ops
    dev = torch.device("cuda:0")
    S, H = 75600, 8
    q, k, v = (torch.randn((H, S, 128), device=dev, dtype=torch.bfloat16) for _ in range(3))
    o = torch.empty_like(q)
    f8 = ops.fp8_quantize_qkv(q, k, v, center_k=True)
    fn = lambda: ops.attn_fwd(f8.q, f8.k, f8.v, o, n_q=S, n_kv=S, v_descale=f8.v_descale)
    fn(); torch.cuda.synchronize()
    rows, stop = [], [False]

    def sample():
        while not stop[0]:
            r = smi()
            if r:
                rows.append(r)
            time.sleep(0.03)
    th = threading.Thread(target=sample, daemon=True); th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 120
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    stop[0] = True; th.join()
    ms = e0.elapsed_time(e1) / n
    rows = rows[len(rows) // 3:]
    clk = sum(r[0] for r in rows) / max(len(rows), 1); pw = sum(r[1] for r in rows) / max(len(rows), 1)
    steps = 10 * 1181  # 9.25 rounds of workgroups per CU -> 10, 1181 key blocks each
    print(f"{tag:22s} {ms:7.3f} ms   sclk {clk:5.0f} MHz   {pw:5.0f} W   ~{ms * 1e-3 * clk * 1e6 / steps:5.0f} cycles per step   ({len(rows)} samples)", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one(sys.argv[1])
    else:
        libs = [("product", os.path.join(ROOT, "vorta_amd", "csrc", "libvorta_hip.so"))] + \
               [(os.path.basename(p)[len("libvorta_hip_d"):-3], p) for p in sorted(glob.glob(os.path.join(ROOT, "vorta_amd", "csrc", "libvorta_hip_d*.so")))]
        for rnd in range(2):
            for tag, lib in libs:
                subprocess.run([sys.executable, os.path.abspath(__file__), tag], env=dict(os.environ, VORTA_HIP_LIB=lib))
