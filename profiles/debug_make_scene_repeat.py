"""Is bench.make_scene deterministic call to call (same seed, same device)?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
ref = None
bits = lambda t: (torch.view_as_real(t) if t.is_complex() else t).contiguous().view(torch.int32)
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    cur = bench.make_scene(2000, 20000, 8000, 0, 20260322, dev)
    junk = [torch.empty((500, 20000), dtype=torch.float64, device=dev).fill_(float(r)) for _ in range(3)]  # allocator traffic as in the bench
    torch.cuda.synchronize()
    if ref is None:
        ref = cur
        continue
    for name, a, b in zip(("inc", "s_vv", "anc"), ref, cur):
        d = (bits(a) != bits(b))
        d = d.any(dim=-1) if d.dim() == 3 else d
        n = int(d.sum().item())
        if n:
            rows = d.any(dim=1).nonzero().flatten()
            print(f"rep {r}: {name}: {n} values differ, lines {rows[:3].tolist()}..{rows[-2:].tolist()}", flush=True)
    del junk
print("done", flush=True)
