"""Does a small chain on a side stream run beside a long chain on the main stream?  (torch only)"""
import sys, time, torch
use_null = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
a = torch.randn(4096, 4096, device=dev); b = torch.randn(4096, 4096, device=dev)
x = torch.zeros(1024, device=dev)
main = torch.cuda.current_stream() if use_null else torch.cuda.Stream(dev)
side = torch.cuda.Stream(dev)
with torch.cuda.stream(main):
    for _ in range(3):
        (a @ b)
torch.cuda.synchronize()
for trial in range(3):
    with torch.cuda.stream(main):
        t0 = time.perf_counter()
        for _ in range(30):
            c = a @ b                       # ~1 ms each
        t_enq = time.perf_counter() - t0
    with torch.cuda.stream(side):
        t1 = time.perf_counter()
        for _ in range(20):
            x.add_(1.0)
        y = x[:4].tolist()                   # D2H + sync of the side stream only
        t_side = time.perf_counter() - t1
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t_rest = time.perf_counter() - t2
    print("main=%s: enqueue %.2f ms, side chain returned after %.2f ms, main needed another %.2f ms" %
          ("null" if use_null else "own", 1e3 * t_enq, 1e3 * t_side, 1e3 * t_rest))
