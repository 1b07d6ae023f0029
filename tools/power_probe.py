"""Sustained (4 s) loops of the one-tile vs persistent GEMM with rocm-smi sampling: burst timing vs power-capped steady state."""
import os, sys, subprocess, threading, time, re, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textflux_amd import ops
BF = torch.bfloat16

def sample(stop, acc):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            sclk = re.findall(r"sclk clock level.*?\((\d+)Mhz\)", r)
            pw = re.findall(r"Power \(W\):\s*([\d.]+)", r)
            acc.append((int(sclk[0]) if sclk else -1, float(pw[0]) if pw else -1))
        except Exception as e:
            acc.append((-1, -1))
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
    stop.set(); th.join(timeout=3)
    a = acc[3:] or acc
    ms, w = dt / n * 1e3, sum(x[1] for x in a) / len(a)
    print(f"{name}: {ms:.3f} ms/launch; sclk {sum(x[0] for x in a) / len(a):.0f} MHz, {w:.0f} W, {ms * w / 1e3:.3f} J/launch ({len(a)} samples)", flush=True)
    return ms, w

for (M, N, K) in ([(36864, 9216, 3072), (36864, 3072, 12288)] if __name__ == "__main__" else []):
    x = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
    b = torch.randn(N, device="cuda").to(BF); out = torch.empty(M, N, dtype=BF, device="cuda")
    xs = (torch.randn(M, K, device="cuda") * 0.05).to(BF)
    for rep in range(2):
        probe(f"{M}x{N}x{K} one-tile", lambda: ops.gemm(x, w, b, out=out, variant=2))
        probe(f"{M}x{N}x{K} persistent", lambda: ops.gemm(x, w, b, out=out, variant=3))
    probe(f"{M}x{N}x{K} hipBLASLt", lambda: torch.nn.functional.linear(x, w, b))
    xz, wz = torch.zeros_like(x), torch.zeros_like(w)
    probe(f"{M}x{N}x{K} one-tile zeros", lambda: ops.gemm(xz, wz, b, out=out, variant=2))
    probe(f"{M}x{N}x{K} persistent zeros", lambda: ops.gemm(xz, wz, b, out=out, variant=3))
