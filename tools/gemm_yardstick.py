"""Measurement only: what the vendor library (rocBLAS / hipBLASLt through torch.mm, f32, no reduced precision) takes for
the six FC GEMM shapes of configs[1] -- a yardstick for kernels_gemm.hip.  python tools/gemm_yardstick.py"""
import torch

torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda", 0)
B, K0, N0, N1 = 4096, 26 * 16 + 13, 512, 256


def bench(name, f, flop, reps=200):
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    print("%-28s %7.2f us  %6.1f TF/s  %.3f of 157.3" % (name, us, flop / us * 1e-6, flop / us * 1e-6 / 157.3))


def r(*s):
    return torch.randn(*s, device=dev, dtype=torch.float32)


for kpad in (K0, 432):
    A0, W0 = r(B, kpad), r(N0, kpad)          # fwd0: A0 [B][K] x W0^T  (both K-contiguous, "NT")
    A1, W1 = r(B, N0), r(N1, N0)
    d1, d0 = r(B, N1), r(B, N0)
    print("K0 = %d" % kpad)
    bench("fwd0  4096x512x%d NT" % kpad, lambda: torch.mm(A0, W0.t()), 2.0 * B * N0 * kpad)
    bench("fwd1  4096x256x512 NT", lambda: torch.mm(A1, W1.t()), 2.0 * B * N1 * N0)
    bench("d1    4096x512x256 NN", lambda: torch.mm(d1, W1), 2.0 * B * N0 * N1)
    bench("d0    4096x%dx512 NN" % kpad, lambda: torch.mm(d0, W0), 2.0 * B * kpad * N0)
    bench("dW1   512x256x4096 TN", lambda: torch.mm(A1.t(), d1), 2.0 * B * N0 * N1)
    bench("dW0   %dx512x4096 TN" % kpad, lambda: torch.mm(A0.t(), d0), 2.0 * B * kpad * N0)
