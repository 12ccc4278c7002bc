import sys, time
sys.path.insert(0, ".")
import torch, alga_amd
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for pile in (0, 1, 0, 1):
    e = alga_amd.Engine(0); e.set_option("pile", pile)
    t = time.perf_counter(); e.reserve(90621096, 144, 82); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("pile", pile, "reserve ms", round(dt * 1e3, 1)); e.close()
for gb in (1, 2, 4, 5.8, 8):
    t = time.perf_counter(); x = torch.empty(int(gb * 2**30), dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("torch.empty", gb, "GB ms", round(dt * 1e3, 1)); del x; torch.cuda.empty_cache()
