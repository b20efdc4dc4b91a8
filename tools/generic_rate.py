import os, sys
sys.path.insert(0, "/root/repo")
import __graft_entry__ as entry
import torch
pkg = entry.load_package(); L = pkg.lib()
"""Developer bench: rows that are not a multiple of 16 bytes — ragged form of the tiled kernel vs the generic kernel."""
for (h, w, c, n, r) in [(1080, 1918, 3, 64, 1), (1080, 1918, 3, 64, 2), (250, 250, 3, 5000, 1), (768, 1366, 3, 64, 1), (768, 1366, 4, 64, 2),
                        (1000, 1000, 1, 64, 1), (1080, 1920, 3, 64, 1)]:
  for rag in (1, 0):
    if (w * c) % 16 == 0 and not rag:
        continue
    L.mi_blur_set_option(b"ragged_tiled", rag)
    a = torch.randint(0, 256, (n, h, w, c), dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2): pkg.check(L.mi_blur_enqueue(a.data_ptr(), b.data_ptr(), w, h, c, r, n, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): pkg.check(L.mi_blur_enqueue(a.data_ptr(), b.data_ptr(), w, h, c, r, n, st))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 10
    print(f"{n}x{w}x{h}x{c} r{r}: {us:9.1f} us  {2*a.numel()/us/1e3:7.1f} GB/s  ({'tiled' if (w*c)%16==0 else ('ragged tiled' if rag else 'generic')})")
