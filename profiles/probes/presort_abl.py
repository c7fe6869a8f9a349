import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    from cachedembedding_amd.functional import presort_window
    from cachedembedding_amd import synthetic
    dev = torch.device("cuda", 0)
    B, F, P, C = 16384, 26, 8, 1779442
    gen = synthetic.SyntheticKJT(synthetic.TABLES["criteo_1tb"], B, 1, "power_law", 0.25, seed=1024, device=dev)
    ids = gen.next_values(P)
    # slots with the window's multiplicity structure: one slot per distinct id
    u, inv = torch.unique(ids.view(-1), return_inverse=True)
    perm = torch.randperm(C, device=dev)[:u.numel()]
    slots = perm[inv].view(P, -1).contiguous()
    off = torch.arange(B * F + 1, dtype=torch.int32, device=dev)
    ko = torch.empty(P, B * F, dtype=torch.int64, device=dev)
    ts = []
    for it in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); presort_window(slots, C, keys_out=ko, offsets=off, include_last_offset=True, hook_features=F); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort(); print("CE_PRESORT_DEBUG=%s" % os.environ.get("CE_PRESORT_DEBUG", "0"), f"{ts[len(ts)//2]*1e3:.1f} us per window of {P} batches")
else:
    for d in (0, 1, 2, 4, 8, 6, 14, 15):
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, CE_PRESORT_DEBUG=str(d)))
