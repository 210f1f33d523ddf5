"""Is the attention loop limited by its schedule or by the clock the chip holds under its power management?  The same
launches on random operands and on all-zero operands (same instruction stream, no operand toggling): dense e4m3 and
dense fp16 attention, with the shader clock from rocm-smi sampled during each."""
import os, sys, json, subprocess, threading, time
sys.path.insert(0, os.getcwd())
import torch
from vorta_amd import ops

def sclk():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        card = next(iter(json.loads(out).values()))
        s = next(v for k, v in card.items() if k.startswith("sclk clock speed"))
        p = next(float(v) for k, v in card.items() if "Power" in k)
        return float(s.strip("()").lower().replace("mhz", "")), p
    except Exception:
        return None, None

class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True); self.rows = []; self.on = False; self.stop = False
    def run(self):
        while not self.stop:
            if self.on:
                self.rows.append(sclk())
            time.sleep(0.03)

dev = torch.device("cuda:0")
S, H = int(os.environ.get("S", 75600)), int(os.environ.get("H", 8))
sm = Sampler(); sm.start()
for kind in ("random", "zeros", "random", "zeros"):
    mk = (lambda: torch.randn((H, S, 128), device=dev, dtype=torch.float16)) if kind == "random" else \
         (lambda: torch.zeros((H, S, 128), device=dev, dtype=torch.float16))
    q, k, v = mk(), mk(), mk()
    o = torch.empty_like(q)
    f8 = ops.fp8_quantize_qkv(q, k, v)
    for name, fn in (("e4m3", lambda: ops.attn_fwd(f8.q, f8.k, f8.v, o, n_q=S, n_kv=S, v_descale=f8.v_descale)),
                     ("fp16", lambda: ops.attn_fwd(q, k, v, o, n_q=S, n_kv=S))):
        fn(); torch.cuda.synchronize()
        sm.rows = []; sm.on = True
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 60 if name == "e4m3" else 30
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        sm.on = False
        ms = e0.elapsed_time(e1) / n
        rows = [r for r in sm.rows if r[0]]
        rows = rows[len(rows) // 3:]
        c = sum(r[0] for r in rows) / max(len(rows), 1); p = sum(r[1] for r in rows) / max(len(rows), 1)
        print(f"{name} dense S={S} H={H} on {kind:6s}: {ms:7.3f} ms  {4.0 * S * S * 128 * H / ms / 1e9:6.0f} TFLOP/s   sclk {c:5.0f} MHz  {p:5.0f} W ({len(rows)} samples)", flush=True)
sm.stop = True
