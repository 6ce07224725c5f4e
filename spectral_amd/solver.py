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

    def solve(self, dbatch, shared, max_iter=0, eps=0.0, out=None):
        """Launches the solve on torch's current stream; returns dict of device tensors."""
        o = out if out is not None else self._buffers(dbatch.B, dbatch.S)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.ctx.solve_device(dbatch.B, dbatch.S, shared, dbatch.seg, dbatch.init, dbatch.ref_end, dbatch.dl_bounds,
                              o["ctrl"], o["cost"], o["status"], o["iters"], stream=stream, max_iter=max_iter, eps=eps)
        return o

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

    def solve_ragged(self, rec, shared, max_iter=0, eps=0.0):
        """Solve a ragged batch record (from corridor_batch); outputs stay on the device."""
        d = self.device
        B, st = rec["B"], rec["seg_stride"]
        o = dict(ctrl=torch.zeros((B, 12 * st), dtype=torch.float64, device=d),
                 cost=torch.empty(B, dtype=torch.float64, device=d),
                 status=torch.empty(B, dtype=torch.int32, device=d), iters=torch.empty(B, dtype=torch.int32, device=d))
        stream = torch.cuda.current_stream(d).cuda_stream
        self.ctx.solve_ragged_device(B, st, shared, rec["seg"], rec["seg_count"], rec["init"], rec["ref_end"],
                                     rec["dl_bounds"], o["ctrl"], o["cost"], o["status"], o["iters"], stream=stream,
                                     max_iter=max_iter, eps=eps)
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
