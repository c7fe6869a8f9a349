"""CLOSED-FORM SGD CHECKER (test infrastructure -- a checker, never the thing measured or shipped).

What plain SGD leaves in an embedding table is known in closed form when no other optimiser state exists
(`torch.optim.SGD(lr)` without momentum / weight decay on the sparse group: recsys/dlrm_main.py:455-461, stepped
at :279 after the backward of :274-278):

    table[row] = table0[row] - lr * sum over every trained lookup j with row(id_j) == row of grad_out[bag(j)]

where row(id) = idx_map[id] and, with one id per bag in feature-major order (recsys/datasets/
criteo.py:127-134), bag(j) = j.  The order of the additions is the only freedom the reference leaves (fp32,
`index_add_`-style coalescing inside a step, steps in sequence), so the check is against the sum accumulated in
fp64, with a per-element bound on what fp32 accumulation in ANY order may differ by.

`SgdLedger` records which batches were trained (references to the id tensors, nothing is copied) and, once the
run has been flushed to the host table, compares every row the run touched -- wherever the cache put it in the
meantime, however often it was evicted and re-admitted -- with that closed form.  It talks to the table through
two callables (initial rows, current rows), so it is independent of the library it checks.  The arithmetic is
torch on the device that holds the ids (sizes: a Criteo-1TB run touches ~10^7 rows of 128 floats).

For the rows that sum the most gradients it also replays the REFERENCE's own arithmetic -- fp32, step by step:
coalesce the step's gradient rows, then w += -lr * g -- and holds it to the same bound: a bound torch's fp32
result violated would not be a statement about the kernels.

Parity note: this is arithmetic that follows from the reference's optimiser configuration, not a golden vector
of the reference; the bag kernels are pinned against torch-CPU elsewhere (tests/test_gpu_bag.py).
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch

# Per-element bound (round 5: BASELINE.json north_star's "within 1e-5 rel fp32", with no absolute floor).
#   n = trained lookups summed into the row, ref64 = the closed form in fp64, S = sum_j |lr * g_j| (per element)
#   n == 1   the row saw ONE fp32 update: it must equal fp32(w0 + fp32(-lr * g)) -- or the fused form
#            fp32(w0 - lr * g), one rounding; both occur (an atomic add of the scaled gradient / an FMA onto the loaded
#            row) -- BIT FOR BIT.  Reported as single_lookup_rows / single_lookup_mismatch; a mismatch is a violation.
#   n <= 4   |got - ref64| <= REL * max(|ref64|, S): relative to the larger of the result and of what was added up
#            (a cold row at |w| ~ 1e-3 is held to ~1e-8, where the round-4 bound allowed 2e-6)
#   n  > 4   |got - ref64| <= REL * |ref64| + SQRT_REL * lr * grad_rms * sqrt(n): fp32 accumulation in ANY order
#            differs by a random walk of roundings whose step follows the size of the summands, so the term is scaled
#            by the workload's gradient (3e-7 * sqrt(n) at the bench's lr = 1, grad_rms = 1e-3: the round-4 constant)
REL, SQRT_REL, COLD_MAX = 1e-5, 3e-4, 4


def elementwise_bound(ref64: torch.Tensor, n: torch.Tensor, abs_sum: torch.Tensor, lr: float, grad_rms: float) -> torch.Tensor:
    """ref64 [k, D] fp64, n [k] lookups per row, abs_sum [k, D] fp64 = sum_j |lr * g_j| (only read where n <= COLD_MAX)
    -> the per-element bound [k, D] described above (the n == 1 rows get the n <= 4 bound here; their bit-for-bit
    check is separate: single_lookup_ok)."""
    nn = n.double().unsqueeze(1)
    cold = torch.maximum(ref64.abs(), abs_sum).mul(REL)
    hot = ref64.abs().mul(REL).add(nn.sqrt() * (SQRT_REL * abs(lr) * grad_rms))
    return torch.where(nn <= COLD_MAX, cold, hot)


def single_lookup_ok(got32: torch.Tensor, w0_32: torch.Tensor, g32: torch.Tensor, lr: float) -> torch.Tensor:
    """[k] bool: got == fp32(w0 + fp32(-lr * g)) or got == fp32(w0 - lr * g) in every element, compared as bits"""
    two = w0_32 + g32 * (-lr)                                                  # fp32 multiply, then fp32 add
    lr32 = float(torch.tensor(-lr, dtype=torch.float32))                      # the scalar as the fp32 arithmetic sees it
    one = (w0_32.double() + g32.double() * lr32).float()                      # fused: exact in fp64, ONE rounding to fp32
    gb = got32.view(torch.int32)
    return ((gb == two.view(torch.int32)) | (gb == one.view(torch.int32))).all(dim=1)


class SgdLedger:
    def __init__(self, num_rows: int, dim: int, lr: float, idx_map: Optional[torch.Tensor] = None):
        """idx_map: id -> table row (int32/int64 [num_ids] on the device) or None for the identity."""
        self.N, self.D, self.lr = int(num_rows), int(dim), float(lr)
        self.idx_map = idx_map
        self.entries: List[tuple] = []            # (ids [n] int64, grad rows [n, D] fp32 in lookup order), training order

    def record(self, ids: torch.Tensor, grad_rows: torch.Tensor) -> None:
        """one trained batch: lookup j read row(ids[j]) and received grad_rows[j]"""
        assert ids.dim() == 1 and grad_rows.shape == (ids.numel(), self.D)
        self.entries.append((ids, grad_rows))

    def __len__(self) -> int:
        return len(self.entries)

    def _rows(self, ids: torch.Tensor) -> torch.Tensor:
        ok = (ids >= 0) & (ids < (self.idx_map.numel() if self.idx_map is not None else self.N))
        safe = torch.where(ok, ids, torch.zeros_like(ids))
        rows = self.idx_map[safe].long() if self.idx_map is not None else safe
        return torch.where(ok, rows, torch.full_like(rows, -1))           # padding / bad ids: no lookup

    @torch.no_grad()
    def check(self, initial_rows: Callable[[torch.Tensor], torch.Tensor],
              current_rows: Callable[[torch.Tensor], torch.Tensor], hot_rows: int = 2048,
              untouched_sample: int = 1 << 20, max_chunk_rows: Optional[int] = None, seed: int = 7) -> dict:
        """initial_rows(rows int64 [k]) / current_rows(rows) -> fp32 [k, D] on the same device.
        Returns counts and the worst cases; `bound_violations == 0` and `untouched_mismatch == 0` is a pass."""
        assert self.entries, "nothing was recorded"
        import time
        dev = self.entries[0][0].device

        def now():
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            return time.time()
        t_start = now()
        N, D, lr = self.N, self.D, self.lr
        # ---- which rows did the run touch, and how many lookups did each get
        mark = torch.zeros(N, dtype=torch.bool, device=dev)
        seen = {}
        for ids, _ in self.entries:
            if id(ids) in seen:
                continue
            seen[id(ids)] = True
            r = self._rows(ids)
            mark[r[r >= 0]] = True
        touched = mark.nonzero(as_tuple=False).view(-1)
        T = int(touched.numel())
        compact = torch.full((N,), -1, dtype=torch.int32, device=dev)
        compact[touched] = torch.arange(T, dtype=torch.int32, device=dev)
        del mark
        cis = {}                                         # id(ids) -> compact index per lookup (-1: no lookup)

        def ci_of(ids):
            """index among the touched rows of every lookup of the batch; T (a dummy row) for padding / bad ids"""
            c = cis.get(id(ids))
            if c is None:
                r = self._rows(ids)
                c = torch.where(r >= 0, compact[r.clamp(min=0)].long(), torch.full_like(r, T))
                cis[id(ids)] = c
            return c

        n_lookups = torch.zeros(T + 1, dtype=torch.int64, device=dev)        # [T] = the dummy row of non-lookups
        for ids, _ in self.entries:
            n_lookups += torch.bincount(ci_of(ids), minlength=T + 1)
        n_lookups = n_lookups[:T]
        g64t = {}

        def grad64t(g):
            v = g64t.get(id(g))
            if v is None:
                v = g64t[id(g)] = g.double().t().contiguous()          # [D, n]: the scan below runs along the last dim
            return v

        def step_sums(ids, g):
            """the step's gradient COALESCED in fp64: (touched-row index [U], sum of the gradient rows of its lookups
            [U, D]).  Sorted + prefix sums along the contiguous dimension rather than index_add_: a Criteo batch sends
            ~9000 lookups at one row, and 9000 fp64 atomics on one address (compare-and-swap loops) cost 100 ms per step;
            a scan along the OUTER dimension is a serial loop per column (170 ms per step)."""
            c = ci_of(ids)
            order = torch.argsort(c)
            uniq, counts = torch.unique_consecutive(c[order], return_counts=True)
            ends = counts.cumsum(0)
            cum = grad64t(g).index_select(1, order).cumsum_(1)
            sums = cum.index_select(1, ends - 1)
            sums[:, 1:] -= cum.index_select(1, ends[:-1] - 1)
            return uniq, sums.t()

        gtensors = {id(g): g for _, g in self.entries}
        cold_cache = {}

        def cold_lookups():
            """the lookups of rows with <= COLD_MAX lookups, gathered ONCE over all entries and grouped by the gradient
            tensor they read: {id(g): (touched-row index [k], position in g [k])} -- a run of thousands of steps has a
            few million of them, and a pass per entry was most of this check's time"""
            if not cold_cache:
                n_ext = torch.cat([n_lookups, torch.full((1,), 1 << 40, dtype=n_lookups.dtype, device=dev)])
                parts = {}
                for ids, g in self.entries:
                    ci = ci_of(ids)
                    pos = (n_ext[ci] <= COLD_MAX).nonzero(as_tuple=False).view(-1)
                    parts.setdefault(id(g), []).append((ci[pos], pos))
                for k_, lst in parts.items():
                    cold_cache[k_] = (torch.cat([a for a, _ in lst]), torch.cat([b for _, b in lst]))
            return cold_cache

        gs = torch.stack([g.float().pow(2).mean() for g in gtensors.values()]).mean().sqrt()
        grad_rms = float(gs)
        free = torch.cuda.mem_get_info(dev)[0] if dev.type == "cuda" else 8 << 30
        chunk = max(1, min(T, int(0.35 * free) // (72 * D)))
        if max_chunk_rows:
            chunk = min(chunk, int(max_chunk_rows))
        res = dict(rows=T, lookups=int(n_lookups.sum()), steps=len(self.entries), bound_violations=0, max_err=0.0,
                   max_err_over_bound=0.0, rows_violating=0, chunks=0, single_lookup_rows=0, single_lookup_mismatch=0,
                   cold_rows=0, max_rel_err_cold=0.0)
        worst = None
        t_marked = now()
        K = min(int(hot_rows), T)
        hot = torch.topk(n_lookups, K).indices if K > 0 else None
        hot_e64 = torch.zeros(K, D, dtype=torch.float64, device=dev) if K > 0 else None
        # ---- every touched row against the fp64 closed form
        for c0 in range(0, T, chunk):
            c1 = min(T, c0 + chunk)
            rows = touched[c0:c1]
            w0 = initial_rows(rows)
            exp = torch.cat([w0.double(), torch.zeros(1, D, dtype=torch.float64, device=dev)])   # + the dummy row
            for ids, g in self.entries:
                uniq, sums = step_sums(ids, g)
                if c0 == 0 and c1 == T:
                    exp.index_add_(0, uniq, sums, alpha=-lr)          # (uniq[-1] may be the dummy row T)
                else:
                    sel = (uniq >= c0) & (uniq < c1)
                    exp.index_add_(0, uniq[sel] - c0, sums[sel], alpha=-lr)
            exp = exp[:c1 - c0]
            if K > 0:
                inch = (hot >= c0) & (hot < c1)
                hot_e64[inch] = exp[hot[inch] - c0]
            # cold rows (n <= COLD_MAX): what was added up, sum_j |lr g_j|, and -- rows with ONE lookup -- that gradient row
            nl = n_lookups[c0:c1]
            is_cold = nl <= COLD_MAX
            cold_of = torch.full((c1 - c0 + 1,), -1, dtype=torch.int64, device=dev)
            n_cold = int(is_cold.sum())
            cold_of[:c1 - c0][is_cold] = torch.arange(n_cold, device=dev)
            s_cold = torch.zeros(n_cold, D, dtype=torch.float64, device=dev)
            one_of = torch.full((c1 - c0 + 1,), -1, dtype=torch.int64, device=dev)
            is_one = nl == 1
            n_one = int(is_one.sum())
            one_of[:c1 - c0][is_one] = torch.arange(n_one, device=dev)
            g_one = torch.zeros(n_one, D, dtype=torch.float32, device=dev)
            for g, (ci_c, pos_c) in cold_lookups().items():             # per distinct gradient tensor: ONE pass
                inch = (ci_c >= c0) & (ci_c < c1)
                ck = cold_of[torch.where(inch, ci_c - c0, torch.full_like(ci_c, c1 - c0))]
                for a in range(0, int(ck.numel()), 1 << 22):            # (bounded temporaries: 4 M lookups x D at a time)
                    ck_a, pos_a = ck[a:a + (1 << 22)], pos_c[a:a + (1 << 22)]
                    sel = ck_a >= 0
                    gs_ = gtensors[g][pos_a[sel]].float()
                    s_cold.index_add_(0, ck_a[sel], gs_.double().abs_(), alpha=abs(lr))
                    ok1 = one_of[ci_c[a:a + (1 << 22)][sel] - c0]
                    s1 = ok1 >= 0
                    g_one[ok1[s1]] = gs_[s1]
            got32 = current_rows(rows)
            if n_one:
                ok = single_lookup_ok(got32[is_one], w0[is_one], g_one, lr)
                res["single_lookup_rows"] += n_one
                res["single_lookup_mismatch"] += int((~ok).sum())
            got = got32.double()
            err = (got - exp).abs_()
            del got, got32
            # (elementwise_bound without a [rows, D] tensor of mostly zeros: the n > 4 form everywhere, then the cold rows)
            bound = exp.abs().mul_(REL).add_(nl.double().sqrt_().mul_(SQRT_REL * abs(lr) * grad_rms).unsqueeze(1))
            if n_cold:
                bound[is_cold] = torch.maximum(exp[is_cold].abs(), s_cold).mul_(REL)
            ratio = torch.where(bound > 0, err / bound.clamp(min=1e-300), (err > 0).double() * float("inf"))
            if n_cold:
                scale = torch.maximum(exp[is_cold].abs(), s_cold)
                rel = torch.where(scale > 0, err[is_cold] / scale.clamp(min=1e-300), (err[is_cold] > 0).double() * float("inf"))
                res["cold_rows"] += n_cold
                res["max_rel_err_cold"] = max(res["max_rel_err_cold"], float(rel.max()))
            bad = ratio > 1.0
            if n_one:                                     # a single-lookup row that is not bit-exact violates, whatever its size
                row_bad = torch.zeros(c1 - c0, dtype=torch.bool, device=dev)
                row_bad[is_one.nonzero(as_tuple=False).view(-1)[~ok]] = True
                bad = bad | (row_bad.unsqueeze(1) & (err > 0))
            res["bound_violations"] += int(bad.sum())
            res["rows_violating"] += int(bad.any(dim=1).sum())
            res["max_err"] = max(res["max_err"], float(err.max()))
            mr = float(ratio.max())
            if mr >= res["max_err_over_bound"]:
                res["max_err_over_bound"] = mr
                k = int(ratio.max(dim=1).values.argmax())
                worst = dict(row=int(rows[k]), lookups=int(n_lookups[c0 + k]), err=float(err[k].max()),
                             ref_abs_max=float(exp[k].abs().max()))
            res["chunks"] += 1
            del err, bound, ratio, bad, exp, w0, s_cold, g_one
        res["worst"] = worst
        res["bound"] = (f"per element: rows with 1 lookup == fp32(w0 - lr*g) bit for bit; <= {COLD_MAX} lookups: |table - ref64| <= "
                        f"{REL:g}*max(|ref64|, sum|lr*g|); more: <= {REL:g}*|ref64| + {SQRT_REL:g}*lr*grad_rms*sqrt(lookups)")
        res["grad_rms"] = grad_rms
        t_main = now()
        # ---- the reference's own fp32 arithmetic on the rows that sum the most gradients, held to the same bound
        if K > 0:
            hot_of = torch.full((T + 1,), -1, dtype=torch.int64, device=dev)
            hot_of[hot] = torch.arange(K, device=dev)
            rows = touched[hot]
            w32 = initial_rows(rows).clone()
            for ids, g in self.entries:                    # training order
                hk = hot_of[ci_of(ids)]
                sel = hk >= 0
                # torch.optim.SGD on a sparse gradient: grad.coalesce() -- the step's gradient rows summed per row in
                # fp32, here with torch's index_add_ (coalesce() itself sorts first: 8 ms per step) -- then ONE
                # param.add_(grad, alpha=-lr) per step
                step = torch.zeros(K, D, dtype=torch.float32, device=dev).index_add_(0, hk[sel], g[sel].float())
                w32.add_(step, alpha=-lr)
            e64 = hot_e64
            # (hot rows sum thousands of gradients; should one have <= COLD_MAX lookups -- a toy run -- it gets the
            # n > 4 form too: abs_sum is not kept per hot row)
            bound = e64.abs() * REL + n_lookups[hot].double().sqrt().mul(SQRT_REL * abs(lr) * grad_rms).unsqueeze(1)
            got = current_rows(rows).double()
            scale = e64.abs().amax(dim=1, keepdim=True).clamp(min=1e-30)
            res["hot_rows_checked"] = K
            res["hot_rows_min_lookups"] = int(n_lookups[hot].min())
            res["hot_rows_max_lookups"] = int(n_lookups[hot].max())
            res["hot_rows_max_abs_value"] = float(e64.abs().max())
            res["hot_torch_fp32_max_err_over_bound"] = float(((w32.double() - e64).abs() / bound).max())
            res["hot_table_max_err_over_bound"] = float(((got - e64).abs() / bound).max())
            res["hot_table_vs_torch_fp32_max_diff_rel_to_row_max"] = float(((got - w32.double()).abs() / scale).max())
        t_hot = now()
        res["seconds_by_part"] = {"mark_and_count": t_marked - t_start, "closed_form_fp64": t_main - t_marked,
                                  "hot_rows_torch_fp32": t_hot - t_main}
        # ---- rows the run never looked up still hold their initial value, bit for bit
        if untouched_sample > 0:
            gen = torch.Generator(device=dev).manual_seed(seed)
            cand = torch.randint(0, N, (int(untouched_sample),), device=dev, generator=gen)
            cand = cand[compact[cand] < 0]
            a, b = initial_rows(cand), current_rows(cand)
            res["untouched_sampled"] = int(cand.numel())
            res["untouched_mismatch"] = int((a.view(torch.int32) != b.view(torch.int32)).any(dim=1).sum())
        return res
