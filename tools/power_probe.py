"""Loop one kernel for a few seconds while sampling rocm-smi (sclk / power) in the background."""
import os, sys, subprocess, threading, time, re, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
BF = torch.bfloat16
M, N, K = 36864, 9216, 3072
x = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
b = torch.randn(N, device="cuda").to(BF); out = torch.empty(M, N, dtype=BF, device="cuda")
D = 3072
y = torch.randn(8, 4608, 3 * D, device="cuda").to(BF); o = torch.empty(8, 4608, D, dtype=BF, device="cuda")
xz, wz = torch.zeros_like(x), torch.zeros_like(w)

def sample(stop, acc):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            sclk = re.findall(r"sclk clock level.*?\((\d+)Mhz\)", r)
            pw = re.findall(r"Power \(W\):\s*([\d.]+)", r)
            acc.append((sclk[:1], pw[:1]))
        except Exception as e:
            acc.append(("err", str(e)[:60]))
        time.sleep(0.3)

def probe(name, fn, secs=4.0):
    stop, acc = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, acc), daemon=True); th.start()
    t0 = time.time(); n = 0
    torch.cuda.synchronize()
    try:
        while time.time() - t0 < secs:
            for _ in range(20):
                fn()
            torch.cuda.synchronize(); n += 20
    except Exception as e:
        stop.set(); print(name, "FAILED", e); return
    dt = time.time() - t0
    stop.set(); th.join()
    print(f"{name}: {dt / n * 1e3:.3f} ms/launch; samples (sclk MHz, W): {acc[2:8]}", flush=True)

probe("gemm8p full", lambda: ops.gemm(x, w, b, out=out, variant=1))
probe("gemm8p no-prefetch", lambda: ops.gemm(x, w, b, out=out, variant=11))
probe("gemm8p mfma-only", lambda: ops.gemm(x, w, b, out=out, variant=17))
probe("gemm8p mfma-only no-setprio", lambda: ops.gemm(x, w, b, out=out, variant=273))
probe("gemm8p full no-setprio", lambda: ops.gemm(x, w, b, out=out, variant=266))
probe("gemm8p mfma-only, zero operands", lambda: ops.gemm(xz, wz, b, out=out, variant=17))
probe("gemm8p no-prefetch, zero operands", lambda: ops.gemm(xz, wz, b, out=out, variant=11))
probe("gemm8p full, zero operands", lambda: ops.gemm(xz, wz, b, out=out, variant=1))
probe("hipBLASLt linear", lambda: torch.nn.functional.linear(x, w, b))
probe("attention", lambda: ops.attention(y[:, :, 2 * D:], y[:, :, :D], y[:, :, D:2 * D], out=o))
