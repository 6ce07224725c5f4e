"""Device-resident batch solver: torch supplies HBM buffers and streams, the HIP library
(libbtrapz_hip.so) does all the work.  No torch ops on the data path."""
import numpy as np
import torch

from . import layout as L
from .native import Context


class DeviceBatch:
    """A candidate batch resident in HBM (field-major SoA, see layout.py)."""

    def __init__(self, batch, device):
        self.B, self.S = batch.B, batch.S
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)
        self.seg, self.init, self.ref_end, self.dl_bounds = f(batch.seg), f(batch.init), f(batch.ref_end), f(batch.dl_bounds)


class BatchSolver:
    """Solves DeviceBatches on one GPU; outputs stay on the device."""

    def __init__(self, device_index=0):
        if not torch.cuda.is_available():
            raise RuntimeError("spectral_amd.BatchSolver needs a HIP device (no CPU path)")
        self.device = torch.device("cuda", device_index)
        self.ctx = Context(device_index)
        self._out = {}

    def upload(self, batch):
        return DeviceBatch(batch, self.device)

    def _buffers(self, B, S):
        key = (B, S)
        if key not in self._out:
            d = self.device
            self._out[key] = dict(ctrl=torch.empty((B, 12 * S), dtype=torch.float64, device=d),
                                  cost=torch.empty(B, dtype=torch.float64, device=d),
                                  status=torch.empty(B, dtype=torch.int32, device=d),
                                  iters=torch.empty(B, dtype=torch.int32, device=d))
        return self._out[key]

    def solve(self, dbatch, shared, max_iter=0, eps=0.0, out=None, warm=None, keep_multipliers=False, elastic=0,
              elastic_tol=0.0, queue=0, split=0, start=0, cap_iter=0, lean=0, compact=0):
        """Launches the solve on torch's current stream; returns dict of device tensors.

        warm: dict with optional "x0" ([B,2,S,3] joint states, e.g. from eval_states) and "lam" ([2,36,B,S]
        multipliers kept by an earlier solve) plus optional "mu0", "smin" and "hint" ([B] int32 expected difficulty,
        e.g. the previous step's iters: scheduling only) -- btrapz_warm.  keep_multipliers
        adds this solve's multipliers to the result as "lam".  elastic: btrapz_options.elastic (0 off, 1 rescue pass
        over stalled candidates, 2 elastic rows for every candidate).  split: btrapz_options.split (0 automatic: the
        one-candidate-per-wavefront form for batches that leave SIMDs idle; 1 always where it applies; -1 never).  lean:
        btrapz_options.lean (the two-wavefronts-per-SIMD form of cold solves: 0 automatic, 1 where it applies, -1 never)."""
        o = out if out is not None else self._buffers(dbatch.B, dbatch.S)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if warm is None and not keep_multipliers:
            self.ctx.solve_device(dbatch.B, dbatch.S, shared, dbatch.seg, dbatch.init, dbatch.ref_end,
                                  dbatch.dl_bounds, o["ctrl"], o["cost"], o["status"], o["iters"], stream=stream,
                                  max_iter=max_iter, eps=eps, elastic=elastic, elastic_tol=elastic_tol, queue=queue,
                                  split=split, start=start, cap_iter=cap_iter, lean=lean, compact=compact)
            return o
        warm = warm or {}
        x0, lam0, hint = warm.get("x0"), warm.get("lam"), warm.get("hint")
        if hint is not None:
            assert hint.dtype == torch.int32 and hint.is_contiguous() and hint.numel() == dbatch.B
        if x0 is not None:
            assert x0.dtype == torch.float64 and x0.is_contiguous() and tuple(x0.shape) == (dbatch.B, 2, dbatch.S, 3)
        if lam0 is not None:
            assert lam0.dtype == torch.float64 and lam0.is_contiguous() and tuple(lam0.shape) == (2, 36, dbatch.B, dbatch.S)
        lam_out = None
        if keep_multipliers:
            # in place when a previous solve's array is passed in (allowed: include/btrapz_hip.h, btrapz_warm)
            lam_out = lam0 if lam0 is not None else torch.empty((2, 36, dbatch.B, dbatch.S), dtype=torch.float64,
                                                                device=self.device)
            o = dict(o); o["lam"] = lam_out
        self.ctx.solve_warm_device(dbatch.B, dbatch.S, shared, dbatch.seg, None, dbatch.init, dbatch.ref_end,
                                   dbatch.dl_bounds, o["ctrl"], o["cost"], o["status"], o["iters"], x0=x0, lam0=lam0,
                                   lam_out=lam_out, mu0=warm.get("mu0", 0.0), smin=warm.get("smin", 0.0),
                                   stream=stream, max_iter=max_iter, eps=eps, hint=hint, elastic=elastic,
                                   elastic_tol=elastic_tol, lean=lean)
        return o

    def prepare(self, dbatch, shared, out=None, **options):
        """solve(dbatch, shared, **options) prepared once: returns (call, out) where call() launches the solve on the
        stream that is current NOW and `out` is the dict of device tensors it fills (cold solves only)."""
        o = out if out is not None else self._buffers(dbatch.B, dbatch.S)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        call = self.ctx.prepared_solve(dbatch.B, dbatch.S, shared, dbatch.seg, dbatch.init, dbatch.ref_end, dbatch.dl_bounds,
                                       o["ctrl"], o["cost"], o["status"], o["iters"], stream=stream, **options)
        return call, o

    def eval_states(self, dbatch, ctrl, times):
        """(p, v, a) of every candidate's solved trajectory at times[b][j] (seconds from the start of its
        horizon) -> [B, 2, n_times, 3]; the x0 of a warm start."""
        times = times.to(self.device, dtype=torch.float64).contiguous()
        n = times.shape[1]
        x = torch.empty((dbatch.B, 2, n, 3), dtype=torch.float64, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.ctx.eval_states_device(dbatch.B, dbatch.S, None, dbatch.seg, ctrl, n, times, x, stream=stream)
        return x

    def corridor_batch(self, kb, variant, seg_stride=16):
        """Device corridor stage on a spectral_amd.knots.KnotBatch -> dict of device tensors forming a ragged
        batch record (seg, seg_count, init, ref_end, dl_bounds)."""
        d = self.device
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(d)
        B = kb.B
        rec = dict(B=B, seg_stride=seg_stride,
                   seg=torch.zeros((L.NUM_SEG_FIELDS, B, seg_stride), dtype=torch.float64, device=d),
                   seg_count=torch.zeros(B, dtype=torch.int32, device=d), init=f(kb.init),
                   ref_end=torch.zeros((B, 2), dtype=torch.float64, device=d),
                   dl_bounds=torch.zeros((B, 10), dtype=torch.float64, device=d))
        ins = [f(kb.s_bounds), f(kb.l_bounds), f(kb.ds_bounds), f(kb.dl_bounds), f(kb.s_ref), f(kb.l_ref)]
        stream = torch.cuda.current_stream(d).cuda_stream
        self.ctx.corridor_batch_device(variant, B, kb.N, kb.num_obs, kb.delta, *ins, seg_stride, rec["seg"],
                                       rec["seg_count"], rec["ref_end"], rec["dl_bounds"], stream=stream)
        rec["_inputs"] = ins  # keep the knot arrays alive until the launch has run
        return rec

    def prism_bounds(self, prisms, N, O, road=None):
        """Obstacle prisms [B, P, 8] (s0, l0, t0, vel_s, vel_l, T, active, -) -> per-knot bounds of the lateral strips
        (btrapz_prism_bounds_device): s_bounds, l_bounds [B, O, N, 2] and n_strips [B], on the device."""
        from .native import CRoad
        d = self.device
        prisms = prisms.to(d, dtype=torch.float64).contiguous()
        B, P = prisms.shape[0], prisms.shape[1]
        sb = torch.empty((B, O, N, 2), dtype=torch.float64, device=d); lb = torch.empty_like(sb)
        n = torch.empty(B, dtype=torch.int32, device=d)
        stream = torch.cuda.current_stream(d).cuda_stream
        self.ctx.prism_bounds_device(B, P, N, road or CRoad.reference(), prisms, O, sb, lb, n, stream=stream)
        return sb, lb, n

    def corridor_batch_tensors(self, variant, N, delta, s_bounds, l_bounds, ds_bounds, dl_bounds, s_ref, l_ref, init,
                               seg_stride=16):
        """corridor_batch on device tensors (e.g. the output of prism_bounds): s_bounds, l_bounds [B, O, N, 2],
        ds_bounds, dl_bounds [B, N, 2], s_ref, l_ref [B, N], init [B, 6]."""
        d = self.device
        B, O = s_bounds.shape[0], s_bounds.shape[1]
        c = lambda t: t.to(d, dtype=torch.float64).contiguous()
        ins = [c(s_bounds), c(l_bounds), c(ds_bounds), c(dl_bounds), c(s_ref), c(l_ref)]
        rec = dict(B=B, seg_stride=seg_stride,
                   seg=torch.zeros((L.NUM_SEG_FIELDS, B, seg_stride), dtype=torch.float64, device=d),
                   seg_count=torch.zeros(B, dtype=torch.int32, device=d), init=c(init),
                   ref_end=torch.zeros((B, 2), dtype=torch.float64, device=d),
                   dl_bounds=torch.zeros((B, 10), dtype=torch.float64, device=d))
        stream = torch.cuda.current_stream(d).cuda_stream
        self.ctx.corridor_batch_device(variant, B, N, O, delta, *ins, seg_stride, rec["seg"], rec["seg_count"],
                                       rec["ref_end"], rec["dl_bounds"], stream=stream)
        rec["_inputs"] = ins
        return rec

    def prism_corridor_batch(self, variant, prisms, N, O, delta, ds_bounds, dl_bounds, s_ref, l_ref, init, seg_stride=16,
                             road=None):
        """prism_bounds + corridor_batch_tensors in one launch (btrapz_prism_corridor_batch_device): the strips are
        evaluated inside the corridor kernel instead of written to memory.  Same record, plus rec["n_strips"]."""
        from .native import CRoad
        d = self.device
        c = lambda t: t.to(d, dtype=torch.float64).contiguous()
        prisms = c(prisms)
        B, P = prisms.shape[0], prisms.shape[1]
        ins = [prisms, c(ds_bounds), c(dl_bounds), c(s_ref), c(l_ref)]
        rec = dict(B=B, seg_stride=seg_stride,
                   seg=torch.zeros((L.NUM_SEG_FIELDS, B, seg_stride), dtype=torch.float64, device=d),
                   seg_count=torch.zeros(B, dtype=torch.int32, device=d), init=c(init),
                   ref_end=torch.zeros((B, 2), dtype=torch.float64, device=d),
                   dl_bounds=torch.zeros((B, 10), dtype=torch.float64, device=d),
                   n_strips=torch.empty(B, dtype=torch.int32, device=d))
        stream = torch.cuda.current_stream(d).cuda_stream
        self.ctx.prism_corridor_batch_device(variant, B, P, N, road or CRoad.reference(), prisms, O, delta, *ins[1:],
                                             seg_stride, rec["seg"], rec["seg_count"], rec["ref_end"], rec["dl_bounds"],
                                             rec["n_strips"], stream=stream)
        rec["_inputs"] = ins
        return rec

    def solve_ragged(self, rec, shared, max_iter=0, eps=0.0, elastic=0, elastic_tol=0.0, cap_iter=0, lean=0, compact=0):
        """Solve a ragged batch record (from corridor_batch); outputs stay on the device.  cap_iter:
        btrapz_options.cap_iter (0 automatic, -1 one launch, n two launches with hand-over after n iterations)."""
        d = self.device
        B, st = rec["B"], rec["seg_stride"]
        o = dict(ctrl=torch.zeros((B, 12 * st), dtype=torch.float64, device=d),
                 cost=torch.empty(B, dtype=torch.float64, device=d),
                 status=torch.empty(B, dtype=torch.int32, device=d), iters=torch.empty(B, dtype=torch.int32, device=d))
        stream = torch.cuda.current_stream(d).cuda_stream
        self.ctx.solve_ragged_device(B, st, shared, rec["seg"], rec["seg_count"], rec["init"], rec["ref_end"],
                                     rec["dl_bounds"], o["ctrl"], o["cost"], o["status"], o["iters"], stream=stream,
                                     max_iter=max_iter, eps=eps, elastic=elastic, elastic_tol=elastic_tol, cap_iter=cap_iter, lean=lean, compact=compact)
        return o

    def argmin(self, cost, group=None, index_base=0):
        """Arg-min of cost over contiguous groups (default: the whole batch). Device tensors."""
        B = cost.numel()
        group = B if group is None else group
        best_idx = torch.empty(B // group, dtype=torch.int64, device=self.device)
        best_cost = torch.empty(B // group, dtype=torch.float64, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.ctx.argmin_device(B, group, index_base, cost, best_idx, best_cost, stream=stream)
        return best_idx, best_cost

    def sample(self, dbatch, ctrl, sel, delta):
        """Bernstein sampling (solve_3d.cc:1279-1392) of the selected candidates."""
        sel = sel.to(self.device, dtype=torch.int64).contiguous()
        t = dbatch.seg[L.F_T]
        max_points = int(torch.floor(t / delta + 1e-9).sum(1).max().item()) + 2
        out = torch.zeros((sel.numel(), 6, max_points), dtype=torch.float64, device=self.device)
        npts = torch.zeros(sel.numel(), dtype=torch.int32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.ctx.sample_device(dbatch.B, dbatch.S, delta, dbatch.seg, dbatch.init, ctrl, sel, max_points, out, npts,
                               stream=stream)
        return out, npts
