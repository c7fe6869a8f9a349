"""GPU parity of the HIP EmbeddingBag kernels (through the C ABI) against the CPU oracle.
Tolerance: fp32 1e-5 relative (BASELINE.json north_star); bit-exact when every bag has one id."""
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = Path(__file__).resolve().parent / "golden"
RTOL, ATOL = 1e-5, 1e-6


def _ce():
    import cachedembedding_amd as ce
    return ce


def _case(seed, N, D, nb, maxlen, minlen=0, weighted=False, off_dtype=torch.int64):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(N, D, generator=g)
    lens = torch.randint(minlen, maxlen + 1, (nb,), generator=g)
    offsets = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(lens, 0)]).to(off_dtype)
    nnz = int(offsets[-1])
    idx = torch.randint(0, N, (nnz,), generator=g)
    psw = torch.rand(nnz, generator=g) if weighted else None
    go = torch.randn(nb, D, generator=g)
    return w, idx, offsets, psw, go


@pytest.mark.parametrize("name,mode,weighted", [("bag_sum", "sum", False), ("bag_sum_weighted", "sum", True),
                                                ("bag_mean", "mean", False)])
def test_golden_forward_backward(name, mode, weighted):
    ce = _ce()
    z = np.load(GOLD / f"{name}.npz")
    w = torch.from_numpy(z["weight"]).cuda().requires_grad_(True)
    idx = torch.from_numpy(z["indices"]).cuda()
    off = torch.from_numpy(z["offsets"]).cuda()
    psw = torch.from_numpy(z["psw"]).cuda() if weighted else None
    out = ce.embedding_bag(idx, w, off, mode=mode, per_sample_weights=psw, include_last_offset=True)
    np.testing.assert_allclose(out.detach().cpu().numpy(), z["out"], rtol=RTOL, atol=ATOL)
    out.backward(torch.from_numpy(z["grad_out"]).cuda())
    np.testing.assert_allclose(w.grad.cpu().numpy(), z["grad_weight"], rtol=RTOL, atol=ATOL)
    # sparse COO gradient form
    w2 = torch.from_numpy(z["weight"]).cuda().requires_grad_(True)
    out2 = ce.embedding_bag(idx, w2, off, mode=mode, per_sample_weights=psw, include_last_offset=True, sparse=True)
    out2.backward(torch.from_numpy(z["grad_out"]).cuda())
    assert w2.grad.is_sparse
    np.testing.assert_allclose(w2.grad.to_dense().cpu().numpy(), z["grad_weight"], rtol=RTOL, atol=ATOL)
    # fused backward+SGD (atomic and deterministic) == oracle SGD.step
    for det in (False, True):
        w3 = torch.from_numpy(z["weight"]).cuda().requires_grad_(True)
        fused = ce.FusedSGD(0.5, deterministic=det)
        out3 = ce.embedding_bag(idx, w3, off, mode=mode, per_sample_weights=psw, include_last_offset=True,
                                sparse=True, fused_sgd=fused)
        out3.backward(torch.from_numpy(z["grad_out"]).cuda())
        assert w3.grad is None
        np.testing.assert_allclose(w3.detach().cpu().numpy(), z["weight_after_sgd"], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("D", [4, 8, 32, 64, 128, 256, 512, 100, 130, 7])
@pytest.mark.parametrize("off_dtype", [torch.int32, torch.int64])
def test_forward_dims_and_ragged(D, off_dtype):
    ce = _ce()
    from oracle import bag_oracle
    w, idx, off, psw, go = _case(D, 513, D, 301, 7, off_dtype=off_dtype)
    out = ce.embedding_bag(idx.cuda(), w.cuda(), off.cuda(), mode="sum", include_last_offset=True)
    ref = bag_oracle.bag_forward(w, idx, off, None, "sum", True)
    torch.testing.assert_close(out.cpu(), ref, rtol=RTOL, atol=ATOL)
    # include_last_offset=False form
    out2 = ce.embedding_bag(idx.cuda(), w.cuda(), off[:-1].cuda(), mode="sum", include_last_offset=False)
    torch.testing.assert_close(out2.cpu(), ref, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("D", [128, 32, 96])
@pytest.mark.parametrize("nb", [1, 63, 64, 65, 4099])
def test_single_id_bags_bit_exact(D, nb):
    """L=1 (every Criteo/Avazu batch): a pure row copy -- must be bit-exact."""
    ce = _ce()
    w, idx, off, _, _ = _case(nb + D, 1000, D, nb, 1, minlen=1, off_dtype=torch.int32)
    out = ce.embedding_bag(idx.cuda(), w.cuda(), off.cuda(), mode="sum", include_last_offset=True)
    assert torch.equal(out.cpu(), w[idx])


@pytest.mark.parametrize("include_last", [True, False])
@pytest.mark.parametrize("F,B,D", [(0, 777, 128), (4, 300, 64), (26, 512, 128), (0, 65, 100)])
def test_forward_without_offsets_is_the_one_id_per_bag_layout(F, B, D, include_last):
    """ce_bag_forward(offsets=NULL) = the same call with offsets = arange (one id per bag): bit-exact, with the shape
    hook, per-sample weights and ignored (-1) indices; NULL with num_bags != nnz is refused.  (Measured: the forward is
    no faster for it -- 55.7-56.4 us either way -- so the Python surface keeps passing the offsets it is given.)"""
    ce = _ce()
    from cachedembedding_amd._lib import check, lib, ptr, stream_ptr
    n = max(F, 1) * B
    g = torch.Generator().manual_seed(n)
    w = torch.randn(3000, D, generator=g).cuda()
    idx = torch.randint(0, 3000, (n,), generator=g).cuda()
    idx[5] = -1
    psw = torch.rand(n, generator=g).cuda()
    off = torch.arange(n + 1 if include_last else n, dtype=torch.int32).cuda()
    for weights in (None, psw):
        outs = []
        for o in (off, None):
            out = torch.full((n, D), 7.0, device="cuda")
            check(lib.ce_bag_forward(ptr(w), 3000, D, ptr(idx), n, ptr(o), 0, n, int(include_last), ptr(weights), 0, F,
                                     ptr(out), stream_ptr()))
            outs.append(out)
        assert torch.equal(outs[0], outs[1])
    assert lib.ce_bag_forward(ptr(w), 3000, D, ptr(idx), n, None, 0, n - 1, int(include_last), None, 0, 0, ptr(out),
                              stream_ptr()) != 0
    ok = idx.clamp(min=0)
    ref = w[ok] * (idx >= 0).unsqueeze(1)
    if F:
        ref = ref.view(F, B, D).transpose(0, 1).contiguous()
    got = ce.embedding_bag(idx, w, off, mode="sum", include_last_offset=include_last, hook_features=F)
    assert torch.equal(got.reshape(ref.shape), ref)
    if n > 10:
        # the same tensor object, no longer arange: bag 2 takes two ids, bag 3 none
        off[3] = 4
        ref2 = ref.clone().view(-1, D) if not F else None
        got2 = ce.embedding_bag(idx, w, off, mode="sum", include_last_offset=include_last, hook_features=0)
        exp = (w[ok] * (idx >= 0).unsqueeze(1)).clone()
        exp[2] = exp[2] + exp[3]
        exp[3] = 0
        torch.testing.assert_close(got2, exp, rtol=0, atol=0)


def test_empty_inputs():
    ce = _ce()
    w = torch.randn(10, 16).cuda()
    # all bags empty
    off = torch.zeros(9, dtype=torch.long).cuda()
    out = ce.embedding_bag(torch.zeros(0, dtype=torch.long).cuda(), w, off, mode="sum", include_last_offset=True)
    assert out.shape == (8, 16) and torch.count_nonzero(out) == 0
    # zero bags
    out = ce.embedding_bag(torch.zeros(0, dtype=torch.long).cuda(), w, torch.zeros(1, dtype=torch.long).cuda(),
                           mode="sum", include_last_offset=True)
    assert out.shape == (0, 16)


def test_2d_input_and_errors():
    ce = _ce()
    from oracle import bag_oracle
    w = torch.randn(50, 32)
    idx = torch.randint(0, 50, (9, 4))
    out = ce.embedding_bag(idx.cuda(), w.cuda(), mode="mean")
    ref = torch.nn.functional.embedding_bag(idx, w, mode="mean")
    torch.testing.assert_close(out.cpu(), ref, rtol=RTOL, atol=ATOL)
    with pytest.raises(ValueError):
        ce.embedding_bag(idx.cuda(), w.cuda(), torch.arange(3).cuda())
    with pytest.raises(NotImplementedError):
        ce.embedding_bag(idx.cuda(), w.cuda(), mode="min")


@pytest.mark.parametrize("F,B,D", [(26, 512, 128), (13, 100, 32), (3, 7, 64)])
def test_shape_hook_fold(F, B, D):
    """hook_features folds sparse_embedding_shape_hook (recsys/models/dlrm.py:26-27) into the store."""
    ce = _ce()
    g = torch.Generator().manual_seed(F)
    w = torch.randn(2000, D, generator=g)
    idx = torch.randint(0, 2000, (F * B,), generator=g)
    off = torch.arange(F * B + 1, dtype=torch.int32)
    wc = w.cuda().requires_grad_(True)
    out = ce.embedding_bag(idx.cuda(), wc, off.cuda(), mode="sum", include_last_offset=True, hook_features=F)
    ref = w[idx].view(F, B, D).transpose(0, 1)
    assert out.shape == (B, F, D) and torch.equal(out.cpu(), ref.contiguous())
    go = torch.randn(B, F, D, generator=g)
    out.backward(go.cuda())
    wr = w.clone().requires_grad_(True)
    torch.nn.functional.embedding_bag(idx, wr, off.long(), mode="sum", include_last_offset=True) \
        .view(F, B, D).transpose(0, 1).backward(go)
    torch.testing.assert_close(wc.grad.cpu(), wr.grad, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("det", [False, True])
def test_fused_sgd_heavy_duplicates(det):
    """Hot rows hit thousands of times in one batch: atomic and sorted updates agree with SGD.step."""
    ce = _ce()
    from oracle import bag_oracle
    g = torch.Generator().manual_seed(9)
    N, D, nb = 64, 128, 8192
    w = torch.randn(N, D, generator=g)
    idx = (torch.rand(nb, generator=g) ** 4 * N).long().clamp_(0, N - 1)
    off = torch.arange(nb + 1, dtype=torch.int32)
    go = torch.randn(nb, D, generator=g) * 0.01
    wc = w.cuda().requires_grad_(True)
    out = ce.embedding_bag(idx.cuda(), wc, off.cuda(), mode="sum", include_last_offset=True, sparse=True,
                           fused_sgd=ce.FusedSGD(1.0, deterministic=det))
    out.backward(go.cuda())
    ref = bag_oracle.sgd_step(w, idx, off, go, 1.0)
    torch.testing.assert_close(wc.detach().cpu(), ref, rtol=1e-4, atol=1e-5)
    if det:
        wc2 = w.cuda().requires_grad_(True)
        out = ce.embedding_bag(idx.cuda(), wc2, off.cuda(), mode="sum", include_last_offset=True, sparse=True,
                               fused_sgd=ce.FusedSGD(1.0, deterministic=True))
        out.backward(go.cuda())
        assert torch.equal(wc2.detach(), wc.detach()), "sorted update must be run-to-run deterministic"


def test_full_size_batch_properties():
    """BASELINE config-3 batch shape (B=16384, F=26, D=128, L=1): checks that do not need the oracle at
    full size -- linearity in the table and the gather identity on a sampled subset."""
    ce = _ce()
    B, F, D, C = 16384, 26, 128, 200_000
    g = torch.Generator(device="cuda").manual_seed(3)
    w = torch.randn(C, D, device="cuda", generator=g)
    idx = torch.randint(0, C, (B * F,), device="cuda", generator=g)
    off = torch.arange(B * F + 1, dtype=torch.int32, device="cuda")
    out = ce.embedding_bag(idx, w, off, mode="sum", include_last_offset=True, hook_features=F)
    out2 = ce.embedding_bag(idx, 2 * w, off, mode="sum", include_last_offset=True, hook_features=F)
    assert torch.equal(out2, 2 * out)
    pick = torch.randint(0, B * F, (4096,), device="cuda", generator=g)
    f, b = pick // B, pick % B
    assert torch.equal(out[b, f], w[idx[pick]])
    assert torch.equal(out.sum(dim=(0, 1)).isfinite().all().cpu(), torch.tensor(True))


@pytest.mark.parametrize("presort", [False, True, "src", "src_identity", "src_excl", "src_excl_shared",
                                     "src_excl_identity"])
def test_full_size_step_vs_torch_cpu(presort):
    """BASELINE config [2] step at its real shape -- cache of C = 1,779,442 rows x 128, B = 16384, F = 26, long-tail
    slots -- against the calls the reference makes on the CPU (F.embedding_bag, then SGD on the summed gradient):
    forward bit-exact (L = 1 is a copy), updated cache rows within 1e-5 relative (fp32 sums in another order).
    presort: False = the backward sorts 1024-lookup tiles itself, True = segment-grouped keys + tile backward,
    "src" = source-row keys + the STREAMING backward (k_bag_presort_seg<true> -> k_bag_bwd_stream);
    "src_identity" = the same keys built with the one-id-per-bag layout stated, so that the FORWARD runs from them too
    (k_bag_fwd_keys) -- the kernel triple bench.py times."""
    ce = _ce()
    from cachedembedding_amd.functional import presort_slots, presort_window
    B, F, D, C, lr = 16384, 26, 128, 1_779_442, 0.5
    g = torch.Generator().manual_seed(11)
    w = torch.randn(C, D, generator=g)
    idx = (torch.rand(B * F, generator=g) ** 6 * C).long().clamp_(0, C - 1)      # hot rows repeat thousands of times
    off = torch.arange(B * F + 1, dtype=torch.int32)
    go = torch.randn(B, F, D, generator=g) * 0.01
    wc = w.cuda().requires_grad_(True)
    if presort in ("src_excl", "src_excl_shared", "src_excl_identity"):
        # owner-exclusive rows (k_bag_presort_seg<true, true> -> k_bag_bwd_stream<EXCL>): ids == slots here.
        # "src_excl": every feature draws from its own slice of the table, so the segments' id ranges are disjoint and
        # the plain-store path is live; "src_excl_shared": all features share the rows (ranges overlap) -> the kernel
        # must notice and keep the atomics
        if presort != "src_excl_shared":
            per = C // F
            idx = (idx % per) + torch.arange(F).repeat_interleave(B) * per
        keys = presort_window(idx.cuda().view(1, -1), C, offsets=off.cuda(), include_last_offset=True,
                              hook_features=F, ids=idx.cuda().view(1, -1),
                              identity_bags=presort == "src_excl_identity")[0]
        # "src_excl_identity" (ADVICE r4): flagged keys AND the forward from the keys -- the flag (bit 31 of the low
        # word) is not part of the output row
        assert keys.identity == (presort == "src_excl_identity")
        assert keys.ranges is not None and keys.ranges.shape == (F, 2)
        flagged = int(((keys.keys & 0x80000000) != 0).sum())
        assert flagged > 1000, "the presort marked no owner-exclusive runs"
        lo, hi = keys.ranges[:, 0].cpu(), keys.ranges[:, 1].cpu()
        disjoint = bool((hi[:-1] < lo[1:]).all())
        assert disjoint == (presort != "src_excl_shared")
    elif presort in ("src", "src_identity"):
        keys = presort_window(idx.cuda().view(1, -1), C, offsets=off.cuda(), include_last_offset=True,
                              hook_features=F, identity_bags=presort == "src_identity")[0]
        assert keys.identity == (presort == "src_identity")
    else:
        keys = presort_slots(idx.cuda(), C) if presort else None
    out = ce.embedding_bag(idx.cuda(), wc, off.cuda(), mode="sum", include_last_offset=True, sparse=True,
                           hook_features=F, fused_sgd=ce.FusedSGD(lr), presorted=keys)
    ref_out = torch.nn.functional.embedding_bag(idx, w, off.long(), mode="sum", include_last_offset=True)
    assert torch.equal(out.detach().cpu(), ref_out.view(F, B, D).transpose(0, 1))
    out.backward(go.cuda())
    gflat = go.transpose(0, 1).reshape(-1, D)
    ref32 = w.clone().index_add_(0, idx, gflat, alpha=-lr)                       # what the reference computes
    ref64 = w.double().index_add_(0, idx, gflat.double(), alpha=-lr)             # what it means
    got = wc.detach().cpu()
    # the bound of oracle/closed_form.py (round 5): rows with ONE lookup bit for bit, rows with <= 4
    # lookups within 1e-5 of max(|ref|, sum |lr g|) -- no absolute floor --, hotter rows 1e-5 relative plus the fp32
    # accumulation noise of n summands of the workload's size (the hottest row here sums ~100 k; torch's own fp32
    # result is held to the same bound)
    from oracle.closed_form import elementwise_bound, single_lookup_ok
    n = torch.bincount(idx, minlength=C)
    abs_sum = torch.zeros(C, D, dtype=torch.float64).index_add_(0, idx, gflat.double().abs(), alpha=lr)
    bound = elementwise_bound(ref64, n, abs_sum, lr, float(gflat.pow(2).mean().sqrt()))
    assert bool(((got.double() - ref64).abs() <= bound).all())
    assert bool(((ref32.double() - ref64).abs() <= bound).all())
    one = (n == 1).nonzero().view(-1)
    assert one.numel() > 10_000
    g_one = torch.zeros(C, D).index_add_(0, idx, gflat)[one]                     # (one summand: exact)
    assert bool(single_lookup_ok(got[one], w[one], g_one, lr).all())             # ONE fp32 update: bit for bit
    cold = ((n <= 4) & (n > 0)).nonzero().view(-1)
    scale = torch.maximum(ref64[cold].abs(), abs_sum[cold])
    assert float(((got[cold].double() - ref32[cold].double()).abs() / scale.clamp(min=1e-300)).max()) <= 1e-5


@pytest.mark.parametrize("nb,F,C,D", [(4096, 4, 3000, 128), (5000, 1, 3000, 128), (1023, 1, 3000, 128), (1, 1, 3000, 128),
                                      (16384, 4, 3000, 64), (16385, 1, 500, 32), (50000, 2, 40000, 128),
                                      (425984, 26, 200000, 32), (70000, 1, 5_000_000, 8)])
def test_presorted_backward_matches_in_kernel_sort(nb, F, C, D):
    """ce_bag_presort (16384-lookup segments) + ce_bag_backward_sgd_presorted == ce_bag_backward_sgd up to fp32
    summation order; the last case has more than 2^22 rows (64-bit tile keys)"""
    ce = _ce()
    from cachedembedding_amd.functional import presort_len, presort_slots
    g = torch.Generator().manual_seed(nb)
    w = torch.randn(C, D, generator=g)
    idx = (torch.rand(nb, generator=g) ** 3 * C).long().clamp_(0, C - 1)
    idx[::17] = C + 5                       # out-of-range slots must be ignored by both paths
    off = torch.arange(nb + 1, dtype=torch.int32)
    go = torch.randn(nb // F, F, D, generator=g) * 0.01 if nb % F == 0 and F > 1 else torch.randn(nb, D, generator=g) * 0.01
    hook = F if go.dim() == 3 else 0
    outs = []
    for use_keys in (False, True):
        wc = w.cuda().requires_grad_(True)
        keys = presort_slots(idx.cuda(), C) if use_keys else None
        if use_keys:
            k = keys.cpu().numpy().view("uint64")
            assert k.size == presort_len(nb) and k.size % 16384 == 0
            for s0 in range(0, k.size, 16384):              # every segment: its valid lookups, equal rows adjacent
                seg = k[s0:s0 + 16384]
                real = seg[seg != 0xFFFFFFFFFFFFFFFF]
                assert (seg[real.size:] == 0xFFFFFFFFFFFFFFFF).all(), "ignored lookups / padding come last"
                lo = idx[s0:min(nb, s0 + 16384)]
                assert real.size == int((lo < C).sum())
                pos = (real & 0xFFFFFFFF).astype("int64")
                assert np.array_equal(np.sort(pos), np.nonzero((lo < C).numpy())[0])       # each lookup once
                rows = (real >> 32).astype("int64")
                assert torch.equal(torch.from_numpy(rows), lo[pos])
                runs = 1 + int((rows[1:] != rows[:-1]).sum()) if rows.size else 0
                assert runs <= 1.35 * np.unique(rows).size + 8, "grouping leaves too many broken runs"
        out = ce.embedding_bag(idx.cuda(), wc, off.cuda(), mode="sum", include_last_offset=True, sparse=True,
                               hook_features=hook, fused_sgd=ce.FusedSGD(0.5), presorted=keys)
        out.backward(go.cuda())
        outs.append(wc.detach().cpu())
    torch.testing.assert_close(outs[0], outs[1], rtol=1e-4, atol=1e-5)
    valid = idx < C
    ref = w.clone()
    gflat = go.transpose(0, 1).reshape(-1, D) if hook else go
    ref.index_add_(0, idx[valid], gflat[valid], alpha=-0.5)
    torch.testing.assert_close(outs[1], ref, rtol=1e-4, atol=1e-5)


def test_presorted_backward_multi_id_bags_mean_and_weights():
    """presorted keys with bags of several ids: mean scaling and per-sample weights follow the lookup, not the slot"""
    ce = _ce()
    from cachedembedding_amd.functional import presort_slots
    g = torch.Generator().manual_seed(9)
    C, D, nb = 700, 64, 9000
    lens = torch.randint(0, 6, (nb,), generator=g)
    off = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(lens, 0)])
    nnz = int(off[-1])
    idx = (torch.rand(nnz, generator=g) ** 2 * C).long().clamp_(0, C - 1)
    w = torch.randn(C, D, generator=g)
    go = torch.randn(nb, D, generator=g) * 0.01
    for mode, psw in (("mean", None), ("sum", torch.rand(nnz, generator=g))):
        wc = w.cuda().requires_grad_(True)
        out = ce.embedding_bag(idx.cuda(), wc, off.cuda(), mode=mode, include_last_offset=True, sparse=True,
                               per_sample_weights=None if psw is None else psw.cuda(), fused_sgd=ce.FusedSGD(0.25),
                               presorted=presort_slots(idx.cuda(), C))
        out.backward(go.cuda())
        ref = w.clone().requires_grad_(True)
        torch.nn.functional.embedding_bag(idx, ref, off, mode=mode, per_sample_weights=psw,
                                          include_last_offset=True).backward(go)
        torch.testing.assert_close(wc.detach().cpu(), w - 0.25 * ref.grad, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("P,n,C", [(3, 16384 * 2, 50000), (4, 26624, 94458), (2, 5000, 300), (8, 16384, 2000)])
def test_presort_window_equals_per_batch_presort_and_feeds_the_backward(P, n, C):
    """one launch for the P batches of a window: batch b's keys cover exactly its own lookups (segments never
    straddle batches, also when a batch is not a whole number of 16384-lookup segments) and drive the fused backward
    to the same result as the in-kernel tile sort"""
    ce = _ce()
    from cachedembedding_amd.functional import presort_len, presort_window
    g = torch.Generator().manual_seed(P * n)
    D = 32
    slots = (torch.rand(P, n, generator=g) ** 3 * C).long().clamp_(0, C - 1)
    slots[:, ::13] = -1                                     # ignored lookups
    keys = presort_window(slots.cuda().contiguous(), C)
    assert tuple(keys.shape) == (P, presort_len(n))
    k = keys.cpu().numpy().view("uint64")
    for b in range(P):
        for s0 in range(0, k.shape[1], 16384):
            seg = k[b, s0:s0 + 16384]
            real = seg[seg != 0xFFFFFFFFFFFFFFFF]
            lo = slots[b, s0:min(n, s0 + 16384)]
            pos = (real & 0xFFFFFFFF).astype("int64")
            assert np.array_equal(np.sort(pos), np.nonzero((lo >= 0).numpy())[0])
            assert torch.equal(torch.from_numpy((real >> 32).astype("int64")), lo[pos])
    w = torch.randn(C, D, generator=g)
    off = torch.arange(n + 1, dtype=torch.int32)
    go = torch.randn(n, D, generator=g) * 0.01
    for b in range(P):
        res = []
        for kk in (None, keys[b]):
            wc = w.cuda().requires_grad_(True)
            ce.embedding_bag(slots[b].cuda(), wc, off.cuda(), mode="sum", include_last_offset=True, sparse=True,
                             fused_sgd=ce.FusedSGD(0.5), presorted=kk).backward(go.cuda())
            res.append(wc.detach().cpu())
        torch.testing.assert_close(res[0], res[1], rtol=1e-4, atol=1e-5)
        valid = slots[b] >= 0
        ref = w.clone().index_add_(0, slots[b][valid], go[valid], alpha=-0.5)
        torch.testing.assert_close(res[1], ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("sparse", [False, True])
def test_padding_idx_and_scale_grad_by_freq(sparse):
    """the two F.embedding_bag arguments the reference forwards that round 1 rejected (sum mode).
    padding_idx is checked against torch-CPU.  scale_grad_by_freq is checked against its documented meaning (every
    row's gradient divided by the number of times the row occurs in the mini-batch): torch's own CPU kernel leaves
    duplicates INSIDE one bag unscaled (probed here: a row looked up twice by the same bag gets the full gradient
    twice), a quirk this path does not copy."""
    ce = _ce()
    g = torch.Generator().manual_seed(21)
    N, D, nb = 300, 48, 500
    lens = torch.randint(0, 5, (nb,), generator=g)
    off = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(lens, 0)])
    nnz = int(off[-1])
    idx = (torch.rand(nnz, generator=g) ** 2 * N).long().clamp_(0, N - 1)
    idx[::7] = 17                                          # the padding id, often
    w0 = torch.randn(N, D, generator=g)
    go = torch.randn(nb, D, generator=g)
    bag = torch.repeat_interleave(torch.arange(nb), lens)
    cnt = torch.bincount(idx, minlength=N).float()
    for kw in (dict(padding_idx=17), dict(scale_grad_by_freq=True), dict(padding_idx=17, scale_grad_by_freq=True)):
        wc = w0.clone().cuda().requires_grad_(True)
        out = ce.embedding_bag(idx.cuda(), wc, off.cuda(), mode="sum", include_last_offset=True, sparse=sparse, **kw)
        out.backward(go.cuda())
        ro = torch.nn.functional.embedding_bag(idx, w0, off, mode="sum", include_last_offset=True,
                                               padding_idx=kw.get("padding_idx"))
        torch.testing.assert_close(out.detach().cpu(), ro, rtol=1e-5, atol=1e-5)
        keep = idx != 17 if "padding_idx" in kw else torch.ones_like(idx, dtype=torch.bool)
        scale = (1.0 / cnt[idx]) if kw.get("scale_grad_by_freq") else torch.ones(nnz)
        want = torch.zeros(N, D).index_add_(0, idx[keep], go[bag[keep]] * scale[keep].unsqueeze(1))
        got = wc.grad.to_dense().cpu() if sparse else wc.grad.cpu()
        torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)
        if "scale_grad_by_freq" not in kw:                 # plain padding: torch agrees
            ref = w0.clone().requires_grad_(True)
            torch.nn.functional.embedding_bag(idx, ref, off, mode="sum", include_last_offset=True, **kw).backward(go)
            torch.testing.assert_close(got, ref.grad, rtol=1e-4, atol=1e-5)
    # the module: padding in ID space, cache op on
    emb = ce.CachedEmbeddingBag(N, D, padding_idx=17, sparse=sparse, _weight=w0.clone(), mode="sum",
                                include_last_offset=True, cuda_row_num=N, warmup_ratio=0.5)
    out = emb(idx.cuda(), off.cuda())
    ro = torch.nn.functional.embedding_bag(idx, w0, off, mode="sum", include_last_offset=True, padding_idx=17)
    torch.testing.assert_close(out.detach().cpu(), ro, rtol=1e-5, atol=1e-5)
    # ... and its gradient (with sparse=True the masked lookups must not reach the COO tensor as index -1):
    # in slot space, compared row by row with torch's gradient in id space
    out.backward(go.cuda())
    gw = emb.cache_weight_mgr.cuda_cached_weight.grad
    if sparse:
        assert int(gw._indices().min()) >= 0
        gw = gw.to_dense()
    ref = w0.clone().requires_grad_(True)
    torch.nn.functional.embedding_bag(idx, ref, off, mode="sum", include_last_offset=True, padding_idx=17).backward(go)
    rows = emb.cache_weight_mgr.cached_idx_map.cpu().long()           # slot -> host row (= id: no frequency map)
    res = rows >= 0
    by_id = torch.zeros(N, D).index_add_(0, rows[res], gw.cpu()[res])
    torch.testing.assert_close(by_id, ref.grad, rtol=1e-4, atol=1e-5)
    # mode='mean' with a padding id: the mean over the NON-padding entries of every bag (torch's rule), forward and
    # gradient, functional and through the module; with scale_grad_by_freq against the closed form
    for kw in (dict(padding_idx=17), dict(scale_grad_by_freq=True), dict(padding_idx=17, scale_grad_by_freq=True)):
        wc = w0.clone().cuda().requires_grad_(True)
        out = ce.embedding_bag(idx.cuda(), wc, off.cuda(), mode="mean", include_last_offset=True, sparse=sparse, **kw)
        out.backward(go.cuda())
        ref = w0.clone().requires_grad_(True)
        ro = torch.nn.functional.embedding_bag(idx, ref, off, mode="mean", include_last_offset=True,
                                               padding_idx=kw.get("padding_idx"))
        torch.testing.assert_close(out.detach().cpu(), ro.detach(), rtol=1e-5, atol=1e-5)
        got = wc.grad.to_dense().cpu() if sparse else wc.grad.cpu()
        keep = idx != 17 if "padding_idx" in kw else torch.ones_like(idx, dtype=torch.bool)
        nvalid = torch.zeros(nb).index_add_(0, bag, keep.float()).clamp_(min=1)
        scale = (1.0 / cnt[idx]) if kw.get("scale_grad_by_freq") else torch.ones(nnz)
        want = torch.zeros(N, D).index_add_(0, idx[keep], go[bag[keep]] * (scale[keep] / nvalid[bag[keep]]).unsqueeze(1))
        torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)
        if "scale_grad_by_freq" not in kw:
            ro.backward(go)
            torch.testing.assert_close(got, ref.grad, rtol=1e-4, atol=1e-5)
    emb_mean = ce.CachedEmbeddingBag(N, D, padding_idx=17, sparse=sparse, _weight=w0.clone(), mode="mean",
                                     include_last_offset=True, cuda_row_num=N)
    torch.testing.assert_close(emb_mean(idx.cuda(), off.cuda()).detach().cpu(),
                               torch.nn.functional.embedding_bag(idx, w0, off, mode="mean", include_last_offset=True,
                                                                 padding_idx=17), rtol=1e-5, atol=1e-5)
    from cachedembedding_amd.functional import presort_slots
    with pytest.raises(NotImplementedError):
        emb(idx.cuda(), off.cuda(), presorted=presort_slots(torch.zeros(nnz, dtype=torch.long, device="cuda"), N))


@pytest.mark.gpu
@pytest.mark.parametrize("D", [4, 20, 32, 128, 260])
@pytest.mark.parametrize("layout", ["hook", "plain", "ragged"])
def test_source_row_keys_backward_matches_tile_backward(D, layout):
    """presort_window(..., offsets=...) -> SrcKeys -> streaming backward (fused SGD and dense) against the unsorted
    backward and a torch index_add_ reference; shared and per-batch offsets, partial last segment, out-of-range rows."""
    import cachedembedding_amd as ce
    from cachedembedding_amd.functional import SrcKeys, presort_window
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(D * 7 + len(layout))
    C, P = 5000, 3
    if layout == "hook":
        B, F = 1500, 13                                   # 19500 lookups: one full + one partial segment
        n = B * F
        offs = torch.arange(n + 1, dtype=torch.int32)
        kw = dict(include_last_offset=True, hook_features=F)
        num_bags = n
    elif layout == "plain":
        n = 20000
        offs = torch.arange(n, dtype=torch.int64)
        kw = dict(include_last_offset=False, hook_features=0)
        num_bags = n
    else:
        num_bags = 3000
        lens = torch.randint(0, 9, (P, num_bags), generator=g)
        lens[:, -1] += 17000 - lens.sum(1)                # equal nnz per batch, ragged bags (some empty)
        assert (lens >= 0).all()
        n = 17000
        offs = torch.cat([torch.zeros(P, 1, dtype=torch.int64), lens.cumsum(1)], 1)
        kw = dict(include_last_offset=True, hook_features=0)
    slots = torch.randint(0, C, (P, n), generator=g)
    hot = torch.randint(0, 8, (P, n), generator=g)         # a few very hot rows: long runs
    slots = torch.where(torch.rand(P, n, generator=g) < 0.3, hot, slots)
    slots[:, 5] = -1                                       # ignored lookups (padding)
    slots[:, 77] = C + 3
    slots = slots.to(dev).contiguous()
    offs_d = offs.to(dev).contiguous()
    keys = presort_window(slots, C, offsets=offs_d, **kw)
    assert isinstance(keys, list) and len(keys) == P and all(isinstance(k, SrcKeys) for k in keys)
    if layout == "hook":
        # one id per bag, in order: stating it (offsets not read by the kernel) gives the same keys, segment by
        # segment (the order inside a bucket is whatever the LDS atomics made it)
        from cachedembedding_amd.functional import is_identity_layout
        assert is_identity_layout(offs_d, True)
        keys_id = presort_window(slots, C, offsets=offs_d, identity_bags=True, **kw)
        for a, b in zip(keys, keys_id):
            ka = a.keys.view(-1, 16384).sort(dim=1).values
            kb = b.keys.view(-1, 16384).sort(dim=1).values
            assert torch.equal(ka, kb)
    out_rows = num_bags
    shape = (num_bags // kw["hook_features"], kw["hook_features"], D) if kw["hook_features"] else (num_bags, D)
    w0 = torch.randn(C, D, generator=g).to(dev)
    for b in range(P):
        ob = offs_d[b] if offs_d.dim() == 2 else offs_d
        go = (torch.randn(*shape, generator=g) * 0.1).to(dev)
        res = {}
        for name, ps in (("src", keys[b]), ("plain", None)):
            w = w0.clone().requires_grad_(True)
            out = ce.embedding_bag(slots[b], w, ob, mode="sum", sparse=True, fused_sgd=ce.FusedSGD(0.5),
                                   presorted=ps, **kw)
            out.backward(go)
            res[name] = w.detach().clone()
            if name == "src":       # dense gradient through the same keys
                w2 = w0.clone().requires_grad_(True)
                ce.embedding_bag(slots[b], w2, ob, mode="sum", presorted=ps, **kw).backward(go)
                w3 = w0.clone().requires_grad_(True)
                ce.embedding_bag(slots[b], w3, ob, mode="sum", **kw).backward(go)
                torch.testing.assert_close(w2.grad, w3.grad, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(res["src"], res["plain"], rtol=1e-5, atol=1e-5)
        # ... and against torch on the CPU: w[slot] -= lr * grad_out[bag of the lookup] for every in-range lookup
        sl = slots[b].cpu()
        obc = ob.cpu().long()
        ends = obc[1:] if kw["include_last_offset"] else torch.cat([obc[1:], torch.tensor([n])])
        bag_of = torch.repeat_interleave(torch.arange(num_bags), ends - obc[:num_bags])
        gsrc = go.cpu()
        gflat = gsrc.transpose(0, 1).reshape(-1, D) if kw["hook_features"] else gsrc.reshape(-1, D)
        ok = (sl >= 0) & (sl < C)
        ref = w0.cpu().double().index_add_(0, sl[ok], gflat[bag_of[ok]].double(), alpha=-0.5)
        torch.testing.assert_close(res["src"].cpu().double(), ref, rtol=1e-5, atol=1e-5)
    # wrong layout / mode are refused, not mis-read
    w = w0.clone().requires_grad_(True)
    ob = offs_d[0] if offs_d.dim() == 2 else offs_d
    with pytest.raises(ValueError):
        ce.embedding_bag(slots[0], w, ob, mode="mean", presorted=keys[0], **kw)
    with pytest.raises(ValueError):
        ce.embedding_bag(slots[0], w, ob, mode="sum", presorted=keys[0]._replace(hook_features=7),
                         include_last_offset=kw["include_last_offset"], hook_features=kw["hook_features"])


@pytest.mark.gpu
@pytest.mark.parametrize("D", [16, 128, 256])
@pytest.mark.parametrize("case", ["disjoint", "shared", "straddle", "hot"])
def test_owner_exclusive_rows_backward(D, case):
    """presort_window(..., ids=...) -> flagged keys + segment id ranges -> fused SGD by plain read-modify-write for
    rows one lane group owns, atomics for the rest, against torch index_add_ in fp64.  disjoint: every 16384-lookup
    segment has its own id range (the exclusive path is live); shared: the segments share ids (must fall back);
    straddle: a feature's lookups cross a segment boundary; hot: a few rows collect thousands of lookups (long runs
    and hot buckets stay on the atomics) next to cold ones."""
    import cachedembedding_amd as ce
    from cachedembedding_amd.functional import presort_window
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(D + len(case))
    P, C = 2, 60000
    if case == "straddle":
        B, F = 10000, 5                                     # 50000 lookups: features cross the 16384 boundaries
    else:
        B, F = 16384, 3
    n = B * F
    per = C // F
    base = torch.arange(F).repeat_interleave(B) * per
    u = torch.rand(P, n, generator=g)
    if case == "hot":
        local = torch.where(u < 0.5, (u * 8).long(), (torch.rand(P, n, generator=g) * per).long())
    else:
        local = (u ** 3 * per).long()
    slots = (local.clamp_(0, per - 1) + base).contiguous()
    if case == "shared":
        slots = (u ** 3 * C).long().clamp_(0, C - 1).contiguous()
    slots[:, 7] = -1                                        # an ignored lookup
    ids = slots.clone()
    ids[:, 7] = 3                                           # its id still counts for the range (conservative)
    offs = torch.arange(n + 1, dtype=torch.int32)
    keys = presort_window(slots.to(dev), C, offsets=offs.to(dev), include_last_offset=True, hook_features=F,
                          ids=ids.to(dev))
    w0 = torch.randn(C, D, generator=g)
    for b in range(P):
        r = keys[b].ranges.cpu()
        nonempty = r[:, 0] <= r[:, 1]
        order = torch.argsort(r[nonempty, 0])
        lo, hi = r[nonempty, 0][order], r[nonempty, 1][order]
        disjoint = bool((hi[:-1] < lo[1:]).all())
        assert disjoint == (case in ("disjoint", "hot")), (case, r)
        go = torch.randn(B, F, D, generator=g) * 0.1
        w = w0.clone().to(dev).requires_grad_(True)
        out = ce.embedding_bag(slots[b].to(dev), w, offs.to(dev), mode="sum", include_last_offset=True, sparse=True,
                               hook_features=F, fused_sgd=ce.FusedSGD(0.5), presorted=keys[b])
        out.backward(go.to(dev))
        ok = slots[b] >= 0
        gflat = go.transpose(0, 1).reshape(-1, D)
        ref = w0.double().index_add_(0, slots[b][ok], gflat[ok].double(), alpha=-0.5)
        cnt = torch.bincount(slots[b][ok], minlength=C).double().unsqueeze(1)
        bound = 1e-5 * ref.abs() + 2e-6 + 3e-7 * cnt.sqrt()
        assert bool(((w.detach().cpu().double() - ref).abs() <= bound).all())
        # the flags do not disturb the entry points that ignore them
        w2 = w0.clone().to(dev).requires_grad_(True)
        ce.embedding_bag(slots[b].to(dev), w2, offs.to(dev), mode="sum", include_last_offset=True,
                         hook_features=F, presorted=keys[b]._replace(ranges=None)).backward(go.to(dev))
        want = torch.zeros(C, D, dtype=torch.float64).index_add_(0, slots[b][ok], gflat[ok].double())
        torch.testing.assert_close(w2.grad.cpu().double(), want, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("D", [8, 128])
def test_hooked_forward_with_ragged_bags_of_total_length_num_bags(D):
    """nnz == num_bags selects the LDS-free forward variant (meant for one id per bag); the bags need not hold one id
    each -- empty and multi-id bags with the same total must take its general path and still land in [B, F, D]."""
    import cachedembedding_amd as ce
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(D)
    B, F, N = 150, 7, 500                                  # 1050 bags: 16 full tiles + a partial one
    nb = B * F
    for ragged in (False, True):
        if ragged:
            lens = torch.ones(nb, dtype=torch.long)
            src = torch.randperm(nb, generator=g)[:300]
            lens[src[:150]] -= 1                           # 150 empty bags ...
            lens[src[150:]] += 1                           # ... 150 bags of two ids: the total stays nb
        else:
            lens = torch.ones(nb, dtype=torch.long)
        off = torch.cat([torch.zeros(1, dtype=torch.long), lens.cumsum(0)])
        idx = torch.randint(0, N, (nb,), generator=g)
        w = torch.randn(N, D, generator=g)
        ref = torch.nn.functional.embedding_bag(idx, w, off[:-1], mode="sum").view(F, B, D).transpose(0, 1)
        wd = w.to(dev).requires_grad_(True)
        out = ce.embedding_bag(idx.to(dev), wd, off.to(dev), mode="sum", include_last_offset=True, hook_features=F)
        torch.testing.assert_close(out.cpu(), ref, rtol=1e-6, atol=1e-6)
        go = torch.randn(B, F, D, generator=g)
        out.backward(go.to(dev))
        wr = w.clone().requires_grad_(True)
        torch.nn.functional.embedding_bag(idx, wr, off[:-1], mode="sum").view(F, B, D).transpose(0, 1).backward(go)
        torch.testing.assert_close(wd.grad.cpu(), wr.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("D", [128, 33, 64])
@pytest.mark.parametrize("fused", [False, True])
def test_mode_max_matches_torch(D, fused):
    """mode='max' (an F.embedding_bag argument the reference forwards, recsys/models/dlrm.py:99-110): forward, the
    gradient routed to the rows that supplied the maxima, and the fused SGD form, against torch on the CPU -- ragged
    bags, an empty bag, a padding index."""
    ce = _ce()
    g = torch.Generator().manual_seed(D)
    R, nb = 500, 200
    w = torch.randn(R, D, generator=g)
    lens = torch.randint(0, 7, (nb,), generator=g)
    lens[3] = 0
    off = torch.cat([torch.zeros(1, dtype=torch.long), lens.cumsum(0)])
    idx = torch.randint(0, R, (int(off[-1]),), generator=g)
    go = torch.randn(nb, D, generator=g)
    wr = w.clone().requires_grad_(True)
    ref = torch.nn.functional.embedding_bag(idx, wr, off, mode="max", include_last_offset=True, padding_idx=7)
    ref.backward(go)
    wc = w.cuda().requires_grad_(True)
    fs = ce.FusedSGD(0.25) if fused else None
    out = ce.embedding_bag(idx.cuda(), wc, off.cuda(), mode="max", include_last_offset=True, padding_idx=7,
                           fused_sgd=fs)
    torch.testing.assert_close(out.cpu(), ref, rtol=0, atol=0)
    out.backward(go.cuda())
    if fused:
        assert wc.grad is None
        torch.testing.assert_close(wc.detach().cpu(), w - 0.25 * wr.grad, rtol=1e-6, atol=1e-6)
    else:
        torch.testing.assert_close(wc.grad.cpu(), wr.grad, rtol=1e-6, atol=1e-6)
    with pytest.raises(RuntimeError):
        ce.embedding_bag(idx.cuda(), wc, off.cuda(), mode="max", include_last_offset=True, sparse=True)
    # folded shape hook: feature-major bags -> [B, F, D]
    F_, B_ = 4, 50
    out_h = ce.embedding_bag(idx.cuda(), w.cuda(), off.cuda(), mode="max", include_last_offset=True, padding_idx=7,
                             hook_features=F_)
    torch.testing.assert_close(out_h.cpu(), ref.detach().view(F_, B_, D).transpose(0, 1).contiguous(), rtol=0, atol=0)


@pytest.mark.parametrize("D", [128, 20])
def test_per_sample_weights_gradient_matches_torch(D):
    ce = _ce()
    g = torch.Generator().manual_seed(D + 1)
    R, nb = 300, 150
    w = torch.randn(R, D, generator=g)
    lens = torch.randint(0, 6, (nb,), generator=g)
    off = torch.cat([torch.zeros(1, dtype=torch.long), lens.cumsum(0)])
    idx = torch.randint(0, R, (int(off[-1]),), generator=g)
    psw = torch.rand(idx.numel(), generator=g)
    go = torch.randn(nb, D, generator=g)
    wr, pr = w.clone().requires_grad_(True), psw.clone().requires_grad_(True)
    torch.nn.functional.embedding_bag(idx, wr, off, mode="sum", include_last_offset=True,
                                      per_sample_weights=pr).backward(go)
    wc, pc = w.cuda().requires_grad_(True), psw.cuda().requires_grad_(True)
    ce.embedding_bag(idx.cuda(), wc, off.cuda(), mode="sum", include_last_offset=True,
                     per_sample_weights=pc).backward(go.cuda())
    torch.testing.assert_close(pc.grad.cpu(), pr.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(wc.grad.cpu(), wr.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("D", [128, 64, 32, 20, 260, 4])
@pytest.mark.parametrize("hook", [True, False])
def test_forward_from_source_row_keys_equals_the_gather_forward(D, hook):
    """ce_bag_forward_src_keys (one id per bag: a cache row loaded once per run of equal rows, stored to every output
    row of the run) against the slot-driven forward and against torch, BIT-exact (L = 1 is a copy); ignored lookups
    (slot -1 / beyond the table) must come out as zero rows, batch sizes that are no multiple of a segment included."""
    ce = _ce()
    from cachedembedding_amd.functional import presort_window
    g = torch.Generator().manual_seed(D * 2 + hook)
    C = 5000
    F, B = (13, 3001) if hook else (1, 40_000)
    n = F * B
    w = torch.randn(C, D, generator=g)
    idx = (torch.rand(n, generator=g) ** 4 * C).long().clamp_(0, C - 1)
    idx[::97] = -1
    idx[5::211] = C + 3
    off = torch.arange(n + 1, dtype=torch.int32)
    keys = presort_window(idx.cuda().view(1, -1), C, offsets=off.cuda(), include_last_offset=True,
                          hook_features=F if hook else 0, identity_bags=True)[0]
    assert keys.identity
    wc = w.cuda()
    kw = dict(mode="sum", include_last_offset=True, hook_features=F if hook else 0)
    by_keys = ce.embedding_bag(idx.cuda(), wc, off.cuda(), presorted=keys, **kw)
    by_slots = ce.embedding_bag(idx.cuda(), wc, off.cuda(), presorted=keys._replace(identity=False), **kw)
    assert torch.equal(by_keys, by_slots)
    safe = idx.clone()
    bad = (idx < 0) | (idx >= C)
    safe[bad] = 0
    ref = w[safe]
    ref[bad] = 0
    if hook:
        ref = ref.view(F, B, D).transpose(0, 1)
    assert torch.equal(by_keys.cpu(), ref)
    # and the backward through the same keys still ignores those lookups
    wg = w.cuda().requires_grad_(True)
    out = ce.embedding_bag(idx.cuda(), wg, off.cuda(), presorted=keys, fused_sgd=ce.FusedSGD(0.5), **kw)
    go = torch.randn(out.shape, generator=g) * 0.01
    out.backward(go.cuda())
    gflat = go.transpose(0, 1).reshape(-1, D) if hook else go
    want = w.double().index_add_(0, safe[~bad], gflat[~bad].double(), alpha=-0.5)
    cnt = torch.bincount(safe[~bad], minlength=C).double().unsqueeze(1)          # row 0 sums thousands of gradients
    bound = 1e-5 * want.abs() + 2e-6 + 3e-7 * cnt.sqrt()
    assert bool(((wg.detach().cpu().double() - want).abs() <= bound).all())


def test_per_sample_weights_gradient_with_fused_sgd_is_refused_before_the_table_moves():
    """the combination used to raise inside backward AFTER the fused kernel had updated the rows."""
    ce = _ce()
    g = torch.Generator().manual_seed(4)
    w = torch.randn(64, 16, generator=g)
    idx = torch.randint(0, 64, (40,), generator=g).cuda()
    off = torch.arange(0, 41, 4).cuda()
    wc = w.cuda().requires_grad_(True)
    pc = torch.rand(40, generator=g).cuda().requires_grad_(True)
    with pytest.raises(NotImplementedError, match="per_sample_weights"):
        ce.embedding_bag(idx, wc, off, mode="sum", include_last_offset=True, per_sample_weights=pc,
                         fused_sgd=ce.FusedSGD(0.5))
    assert torch.equal(wc.detach().cpu(), w)
    # weights that need no gradient are fine with the fused update
    out = ce.embedding_bag(idx, wc, off, mode="sum", include_last_offset=True, per_sample_weights=pc.detach(),
                           fused_sgd=ce.FusedSGD(0.5))
    out.backward(torch.ones_like(out))
    assert not torch.equal(wc.detach().cpu(), w)


def test_mode_max_treats_nan_like_torch():
    """ADVICE r3 asked for NaN to win every comparison "as torch does".  torch's CPU kernel (the oracle here) does not:
    it takes the bag's first row as it is and a later row only if `v > best`, so a NaN in the FIRST row of a bag stays
    and a NaN further on never wins.  The kernel keeps exactly that (bit-parity with the reference path is the
    contract); this pins it, forward values and gradient routing."""
    ce = _ce()
    g = torch.Generator().manual_seed(2)
    R, D, nb = 50, 24, 12
    w = torch.randn(R, D, generator=g)
    w[7, 3] = float("nan")
    w[9] = float("nan")
    idx = torch.randint(0, R, (nb * 4,), generator=g)
    idx[(idx == 7) | (idx == 9)] = 11
    idx[0:4] = torch.tensor([1, 7, 2, 3])        # NaN in the middle of a bag: never wins
    idx[4:8] = torch.tensor([9, 9, 9, 9])        # nothing but NaN rows
    idx[8:12] = torch.tensor([7, 4, 9, 5])       # NaN first: stays
    idx[12:16] = torch.tensor([4, 5, 9, 6])      # a whole NaN row in the middle
    off = torch.arange(0, nb * 4 + 1, 4)
    wr = w.clone().requires_grad_(True)
    ref = torch.nn.functional.embedding_bag(idx, wr, off, mode="max", include_last_offset=True)
    wc = w.cuda().requires_grad_(True)
    out = ce.embedding_bag(idx.cuda(), wc, off.cuda(), mode="max", include_last_offset=True)
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=0, atol=0, equal_nan=True)
    assert bool(torch.isnan(ref[1]).all()) and bool(torch.isnan(ref[2, 3])) and not bool(torch.isnan(ref[0]).any())
    go = torch.randn(nb, D, generator=g)
    ref.backward(go)
    out.backward(go.cuda())
    torch.testing.assert_close(wc.grad.cpu(), wr.grad, rtol=1e-6, atol=1e-6)


def test_coalesced_sparse_backward_of_two_same_sized_tables_on_two_streams():
    """the dedupe scratch of the coalesced COO gradient was a module-global keyed by (device, rows), shared
    by every same-sized table whatever stream its backward ran on."""
    ce = _ce()
    g = torch.Generator().manual_seed(8)
    R, D, n = 20_000, 32, 60_000
    off = torch.arange(n + 1, device="cuda")
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    tabs, idxs, gos, outs = [], [], [], []
    for k in range(2):
        tabs.append(torch.randn(R, D, generator=g).cuda().requires_grad_(True))
        idxs.append((torch.rand(n, generator=g) ** 3 * R).long().clamp_(0, R - 1).cuda())
        gos.append((torch.randn(n, D, generator=g) * 0.01).cuda())       # (row 0 sums ~2000 of them in fp32)
    torch.cuda.synchronize()
    for rep in range(5):
        for k in range(2):
            tabs[k].grad = None
            with torch.cuda.stream(streams[k]):
                out = ce.embedding_bag(idxs[k], tabs[k], off, mode="sum", include_last_offset=True, sparse=True)
                out.backward(gos[k])
        torch.cuda.synchronize()
        for k in range(2):
            gw = tabs[k].grad
            assert gw.is_sparse
            ref = torch.zeros(R, D, device="cuda").index_add_(0, idxs[k], gos[k])
            torch.testing.assert_close(gw.to_dense(), ref, rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("norm_type", [2.0, 1.0, 3.0, float("inf")])
@pytest.mark.parametrize("D", [128, 17])
def test_max_norm_renormalises_the_named_rows_like_torch(D, norm_type):
    """max_norm / norm_type: the rows the input names are scaled in place before the lookup (torch.embedding_renorm_),
    once however often they are named; all other rows stay as they are."""
    ce = _ce()
    g = torch.Generator().manual_seed(int(D * 10 + min(norm_type, 9)))
    R, nb = 400, 120
    w = torch.randn(R, D, generator=g) * 0.3
    idx = torch.randint(0, R // 2, (nb * 3,), generator=g)          # many duplicates, half of the table untouched
    off = torch.arange(0, nb * 3 + 1, 3)
    max_norm = float(torch.linalg.vector_norm(w[:R // 2], ord=norm_type, dim=1).median())   # about half the rows shrink
    wr = w.clone()
    ref = torch.nn.functional.embedding_bag(idx, wr, off, mode="sum", include_last_offset=True, max_norm=max_norm,
                                            norm_type=norm_type)
    wc = w.cuda()
    out = ce.embedding_bag(idx.cuda(), wc, off.cuda(), mode="sum", include_last_offset=True, max_norm=max_norm,
                           norm_type=norm_type)
    assert not torch.equal(wr, w), "the case must renormalise something"
    torch.testing.assert_close(wc.cpu(), wr, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-5)
    assert torch.equal(wc.cpu()[R // 2:], w[R // 2:])


@pytest.mark.parametrize("hook", [True, False])
def test_out_argument_and_the_buffer_picker(hook):
    """embedding_bag(out=...) writes the pooled output into the caller's tensor (a static buffer for graph-captured
    steps) with the same values and the same backward; functional.pick_fast_buffer hands out a tensor of the asked
    shape together with what it measured on every candidate (ce_probe_rows)."""
    ce = _ce()
    from cachedembedding_amd.functional import pick_fast_buffer, probe_rows
    B, F, D, C = 512, 6, 128, 4000
    g = torch.Generator().manual_seed(5)
    w = torch.randn(C, D, generator=g)
    idx = torch.randint(0, C, (B * F,), generator=g).cuda()
    off = torch.arange(B * F + 1, dtype=torch.int32).cuda()
    shape = (B, F, D) if hook else (B * F, D)
    buf, rep = pick_fast_buffer(shape, torch.device("cuda"), F, candidates=3, use="write")
    assert tuple(buf.shape) == shape and buf.is_contiguous() and len(rep["us"]) == 3 and 0 <= rep["picked"] < 3
    assert rep["us"][rep["picked"]] == min(rep["us"]) and min(rep["us"]) > 0
    wr, rd = probe_rows(torch.empty(shape, device="cuda"), F)
    assert wr > 0 and rd > 0
    kw = dict(mode="sum", include_last_offset=True, hook_features=F if hook else 0)
    # work=: the candidates are timed on the caller's own launches (a hipGraph of `reps` calls) and all of them are tried;
    # "read" candidates arrive zero-filled
    wk = w.cuda()
    seen = []

    def fwd(b):
        seen.append(b.data_ptr())
        with torch.no_grad():
            ce.embedding_bag(idx, wk, off, out=b, **kw)

    buf2, rep2 = pick_fast_buffer(shape, torch.device("cuda"), F, candidates=3, use="write", work=fwd, reps=4)
    assert len(rep2["us"]) == 3 and len(set(seen)) == 3 and buf2.data_ptr() in seen and min(rep2["us"]) > 0
    assert rep2["us"][rep2["picked"]] == min(rep2["us"]) and len(rep2["new_segment"]) == 3
    torch.cuda.synchronize()
    assert torch.equal(buf2, ce.embedding_bag(idx, wk, off, **kw))
    zeros = []
    buf3, _ = pick_fast_buffer(shape, torch.device("cuda"), F, candidates=2, use="read", reps=2,
                               work=lambda b: zeros.append(b) or b.add_(0.0))
    torch.cuda.synchronize()
    assert all(float(z.abs().max()) == 0.0 for z in zeros)
    w1 = w.cuda().requires_grad_(True)
    ref = ce.embedding_bag(idx, w1, off, **kw)
    w2 = w.cuda().requires_grad_(True)
    got = ce.embedding_bag(idx, w2, off, out=buf, **kw)
    assert got.data_ptr() == buf.data_ptr() and torch.equal(got, ref)
    go = torch.randn(shape, generator=g).cuda()
    ref.backward(go)
    got.backward(go)
    # (the dense backward folds a row's lookups with fp32 atomics: rows with three or more lookups may differ by an ulp
    # or two between two launches -- tolerance 1e-6 absolute on gradients of size ~1)
    torch.testing.assert_close(w1.grad, w2.grad, rtol=0, atol=1e-6)
    with pytest.raises(ValueError):
        ce.embedding_bag(idx, w2, off, out=torch.empty(3, 3, device="cuda"), **kw)
