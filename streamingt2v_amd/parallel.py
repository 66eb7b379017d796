"""Process-group plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm; "gloo" in CPU tests).

What shards in StreamingT2V:
  * independent videos                      -> replicas, no data-path collective (bench.py --gpus N, "weak" scaling);
  * the two classifier-free-guidance halves of one chunk -> `CfgPairExchange`: ranks 2k / 2k+1 each run ONE half of the
    CFG batch through StreamingWrapper and exchange the raw network outputs with one all-gather per Euler step
    (25 x 72 x 128 x 4 fp32 = 3.7 MB); exact, because nothing in the forward couples the two halves
    (per-sample GroupNorm statistics, per-sample attention, ControlNet batch splits 7 + 7);
  * the enhancement stage's blending windows -> streamingt2v_amd/blending.py.
The AR chunk loop itself is sequential (chunk k+1 needs decoded frames of chunk k).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, device=None):
    """Initialise the default group from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run sets them)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 or dist.is_initialized():
        return world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = torch.device(device)
    dist.init_process_group(backend, **kw)
    return world


# CONTROL PLANE.  A host-side gloo group over all ranks, next to the data-path backend: the verdict of the job-plan preflight travels over it
# (with a timeout), and when that preflight found the data-path backend HUNG inside a collective -- which cannot be cancelled or caught -- the
# bench's barrier and max-over-ranks move onto it too ("use"), so that the replica fallback still produces its number instead of hanging.
_CONTROL = {"group": None, "use": False}


def control_group(timeout_s=120.0):
    """Created once, by every rank, in the same place of the program (JobPlan.__init__ with preflight)."""
    if _CONTROL["group"] is None:
        import datetime
        _CONTROL["group"] = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=timeout_s))
    return _CONTROL["group"]


def max_over_ranks(seconds, device="cpu"):
    """Job time = slowest rank (bench contract)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    if _CONTROL["use"]:
        t = torch.tensor([seconds], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_CONTROL["group"])
        return t.item()
    t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        if _CONTROL["use"]:
            dist.barrier(group=_CONTROL["group"])
        else:
            dist.barrier()


# ---- collectives ---------------------------------------------------------------------------------------------------------------
# Thin wrappers over torch.distributed used by every data-path exchange of the package.  With the "nccl" backend (= RCCL over xGMI on
# ROCm) they are the plain collectives on device tensors.  With "gloo" and DEVICE tensors the payload is staged through host memory:
# that is how the multi-process GPU tests run several ranks of the real HIP path on ONE leased GPU (RCCL refuses two ranks on the same
# device) -- tests/test_gpu_multiproc.py.  Results are bit-identical either way (pure data movement; the fp64 sum of two numbers commutes).
def _staged(t, group):
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_gather(parts, x, group=None):
    x = x.contiguous()
    if _staged(x, group):
        hp = [torch.empty(p.shape, dtype=p.dtype) for p in parts]
        dist.all_gather(hp, x.cpu(), group=group)
        for p, h in zip(parts, hp):
            p.copy_(h)
    else:
        dist.all_gather(parts, x, group=group)
    return parts


def all_to_all_single(recv, send, recv_splits, send_splits, group=None):
    send = send.contiguous()
    if _staged(send, group):
        h = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(h, send.cpu(), recv_splits, send_splits, group=group)
        recv.copy_(h)
    else:
        dist.all_to_all_single(recv, send, recv_splits, send_splits, group=group)
    return recv


def all_reduce_sum(t, group=None):
    if _staged(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def broadcast(t, src, group=None):
    """src: GLOBAL rank.  t must be contiguous (a slice along dim 0 of a contiguous tensor is)."""
    if _staged(t, group):
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)
    return t


def shard_items(n_items, rank, world):
    """Indices of the independent work items (videos) a rank owns: round-robin."""
    return list(range(rank, n_items, world))


class CfgPairExchange:
    """All-gather of the two CFG halves' network outputs inside a pair of ranks.

    rank parity selects the half (even: unconditional, odd: conditional).  `gather(net_half)` returns the concatenated
    [uncond | cond] tensor on both ranks, i.e. exactly what a single process computes for the CFG batch of 2."""

    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        assert dist.get_world_size(group) == 2, "a CFG pair has exactly two ranks"

    @property
    def half(self):
        return self.rank            # 0: uncond half, 1: cond half

    def gather(self, net_half):
        parts = [torch.empty_like(net_half), torch.empty_like(net_half)]
        all_gather(parts, net_half, group=self.group)
        return torch.cat(parts, 0)


def _permute_rows(x, dims, perm):
    """Rows [d0 d1 d2 d3, C] -> rows ordered by `perm` (the repack either side of the sequence-parallel all-to-all): ONE HIP kernel pass
    (svd_permute_rows) instead of torch's generic strided copy."""
    from . import ops
    return ops.permute_rows(x.contiguous(), dims, perm)


def split_sizes(n, parts):
    """Contiguous near-even split of n items into `parts` ranges: the first n % parts ranges get one more (25 frames / 4 -> 7 6 6 6)."""
    q, r = divmod(n, parts)
    return [q + (1 if i < r else 0) for i in range(parts)]


class SeqParallel:
    """Frame <-> pixel sequence parallelism inside ONE forward of the denoiser (SURVEY.md 8e option 2).

    Every spatial operator of the UNet / ControlNet (2-D convolutions, spatial attention, per-frame GroupNorm, LayerNorm, feed-forward)
    is independent per FRAME; every temporal operator (temporal attention, (3,1,1) convolutions, the CAM merger attention) is
    independent per PIXEL.  Rank r of the group therefore holds a contiguous range of the T frames of each batch element for the spatial
    operators ("frame layout": rows (b, t_local, pixel)), and a contiguous range of the pixels of ALL frames for the temporal ones
    ("pixel layout": rows (b, t, pixel_local)); `to_pixels` / `to_frames` move a token matrix between the two with ONE all-to-all
    (RCCL over xGMI; every rank exchanges 1/S^2 of the tensor with every other rank, all links busy).  The only reductions are the
    5-D GroupNorm statistics, which pool over frames AND pixels: `allreduce_sums` adds the per-rank (sum, sum of squares) pairs
    (2 x 32 groups x batch doubles).
    """

    def __init__(self, group=None):
        self.group = group
        self.size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    # ---- index bookkeeping -------------------------------------------------------------------------------------------------------
    def frame_counts(self, T):
        return split_sizes(T, self.size)

    def frame_range(self, T):
        c = self.frame_counts(T)
        lo = sum(c[:self.rank])
        return lo, lo + c[self.rank]

    def pix_local(self, pix):
        assert pix % self.size == 0, f"{pix} pixels do not split over {self.size} ranks"
        return pix // self.size

    def take_frames(self, v, B, T, per=1):
        """rows (b, t, per) of a [B*T*per, ...] tensor -> this rank's frames (b, t_local, per)."""
        lo, hi = self.frame_range(T)
        return v.reshape(B, T, per, *v.shape[1:])[:, lo:hi].reshape(B * (hi - lo) * per, *v.shape[1:]).contiguous()

    # ---- layout changes ----------------------------------------------------------------------------------------------------------
    def to_pixels(self, x, B, T, pix):
        """frame layout [B * Tl * pix, C] -> pixel layout [B * T * pixl, C]."""
        S, C = self.size, x.shape[1]
        cnt, pl = self.frame_counts(T), self.pix_local(pix)
        tl = cnt[self.rank]
        send = _permute_rows(x, (B, tl, S, pl), (2, 0, 1, 3))                              # [dest][b][t_local][pixel_local][c], one pass
        recv = torch.empty((B * T * pl, C), dtype=x.dtype, device=x.device)
        all_to_all_single(recv, send, [B * c * pl for c in cnt], [B * tl * pl] * S, group=self.group)
        if B == 1:
            return recv                                                                   # source order == frame order
        parts = recv.split([B * c * pl for c in cnt], 0)
        return torch.cat([p.reshape(B, c, pl, C) for p, c in zip(parts, cnt)], 1).reshape(B * T * pl, C)

    def to_frames(self, x, B, T, pix):
        """pixel layout [B * T * pixl, C] -> frame layout [B * Tl * pix, C] (inverse of to_pixels)."""
        S, C = self.size, x.shape[1]
        cnt, pl = self.frame_counts(T), self.pix_local(pix)
        tl = cnt[self.rank]
        if B == 1:
            send = x
        else:
            xs = x.reshape(B, T, pl, C).split(cnt, 1)
            send = torch.cat([p.reshape(-1, C) for p in xs], 0)
        recv = torch.empty((S * B * tl * pl, C), dtype=x.dtype, device=x.device)          # [source = pixel range][b][t_local][pixel_local]
        all_to_all_single(recv, send, [B * tl * pl] * S, [B * c * pl for c in cnt], group=self.group)
        return _permute_rows(recv, (S, B, tl, pl), (1, 2, 0, 3))                          # [b][t_local][pixel range][pixel_local] = frame layout

    # ---- small collectives ---------------------------------------------------------------------------------------------------------
    def allreduce_sums(self, sums):
        return all_reduce_sum(sums, group=self.group)

    def gather_frames(self, x, B, T, per):
        """frame layout [B * Tl * per, C] of every rank -> all frames [B * T * per, C] on every rank (uneven ranges are padded to the
        largest one for the collective: NCCL / RCCL all-gather wants equal sizes)."""
        cnt = self.frame_counts(T)
        tmax, C = max(cnt), x.shape[1]
        tl = cnt[self.rank]
        buf = x.reshape(B, tl * per, C)
        if tl < tmax:
            buf = torch.cat([buf, buf.new_zeros(B, (tmax - tl) * per, C)], 1)
        parts = [torch.empty_like(buf) for _ in range(self.size)]
        all_gather(parts, buf, group=self.group)
        return torch.cat([p[:, : c * per] for p, c in zip(parts, cnt)], 1).reshape(B * T * per, C)


class JobPlan:
    """How the ranks of one node share ONE stage-1 job (bench.py --parallelism job) -- or do not (replica).

    world = 1: everything local.  world = 2: the CFG pair (ranks 0 | 1 evaluate the unconditional | conditional half, one all-gather
    of the 3.7 MB network output per Euler step).  world = 4 / 8: CFG pair x sequence parallelism of degree world / 2 inside each half:
    rank = 2 * sp_rank + cfg_half, i.e. the SP group of a half is {half, half + 2, half + 4, ...}.  The temporal-VAE decode of a chunk
    is sharded by its independent 8-frame groups over all ranks (broadcast of each group's frames from its owner).
    Amdahl: the Euler update, conditioning glue and the per-chunk hand-over are replicated (< 1 % of a chunk)."""

    def __init__(self, world=1, rank=0, mode="job", frames_cond=7, min_pix=144, preflight=True):
        """frames_cond: conditioning frames of the ControlNet (the sequence-parallel degree cannot exceed it: every rank needs at least one);
        min_pix: pixels of the LOWEST UNet level (9 x 16 at the 72 x 128 latent; the pixel layout splits them evenly).  A plan that the
        shapes cannot carry is refused HERE, with the reason, instead of as an assert deep inside a forward.  preflight: run one tiny
        instance of every collective the plan uses (uneven all-to-all, padded all-gather, fp64 all-reduce, broadcast) before any model is
        built; a backend that cannot do them makes the plan fall back to replicas (`fallback_reason` says why) instead of dying mid-job."""
        self.world, self.rank, self.mode = world, rank, mode
        self.cfg_exchange, self.sp, self.n_videos, self.video_id = None, None, 1, 0
        self.decode_group = None
        self.fallback_reason = None
        self.abort_report = None          # set when a preflight collective hung: what happened to the data-path communicators
        if world == 1:
            return
        if mode == "job":
            why = self.validate(world, frames_cond, min_pix)
            if why is not None:
                self.mode, self.fallback_reason = "replica", why
        if self.mode == "pairs" and world % 2:
            self.mode, self.fallback_reason = "replica", f"CFG pairs need an even number of GPUs, got {world}"
        if self.mode == "replica":
            self.n_videos, self.video_id = world, rank
            return
        if self.mode == "pairs":
            # world / 2 independent videos, each on a CFG pair: ranks (2k, 2k+1) evaluate the unconditional | conditional half of every network
            # call of video k and all-gather the 3.7 MB network output once per Euler step; the pair also splits the decode's frame groups.
            # No sequence parallelism, no all-to-all: the only collectives are a 2-rank all-gather and a 2-rank broadcast.
            pairs = [dist.new_group([2 * k, 2 * k + 1]) for k in range(world // 2)]
            self.cfg_exchange = CfgPairExchange(pairs[rank // 2])
            self.decode_group = pairs[rank // 2]
            self.n_videos, self.video_id = world // 2, rank // 2
            if preflight:
                why = self._preflight()
                if why is not None:
                    self.mode, self.fallback_reason = "replica", why
                    self.cfg_exchange = self.decode_group = None
                    self.n_videos, self.video_id = world, rank
            return
        pairs = [dist.new_group([2 * k, 2 * k + 1]) for k in range(world // 2)]           # every rank creates every group, same order
        halves = [dist.new_group(list(range(h, world, 2))) for h in (0, 1)] if world > 2 else [None, None]
        self.cfg_exchange = CfgPairExchange(pairs[rank // 2])
        if world > 2:
            self.sp = SeqParallel(halves[rank % 2])
        self.decode_group = dist.group.WORLD
        if preflight:
            why = self._preflight()
            if why is not None:          # every rank reaches the same verdict (the check ends in an all-reduce of the failure flags)
                self.mode, self.fallback_reason = "replica", why
                self.cfg_exchange = self.sp = self.decode_group = None
                self.n_videos, self.video_id = world, rank

    @staticmethod
    def validate(world, frames_cond=7, min_pix=144):
        """None if `world` ranks can share one job, else the reason (shown in the bench line)."""
        if world % 2:
            return f"job parallelism needs an even number of GPUs (CFG pair x sequence parallelism), got {world}"
        sp = world // 2
        if sp > frames_cond:
            return f"sequence-parallel degree {sp} exceeds the {frames_cond} conditioning frames of the ControlNet (max {2 * frames_cond} GPUs)"
        if sp > 1 and min_pix % sp:
            return f"the lowest UNet level has {min_pix} pixels per frame, not divisible by the sequence-parallel degree {sp}"
        return None

    def _preflight(self, timeout_s=None):
        """One tiny instance of every collective of the plan on the device the job will use.  Returns None or the failure text.

        The collectives run on a helper thread that is joined with a timeout (SVD_PREFLIGHT_TIMEOUT, default 90 s): a backend that HANGS inside
        a collective -- which neither raises nor can be cancelled -- is reported like one that throws, and moves the bench's own barrier /
        max-over-ranks onto the gloo control group (`_CONTROL["use"]`).  The verdict is exchanged over that control group, whose own timeout
        bounds the wait for a rank that died."""
        import threading
        timeout_s = float(os.environ.get("SVD_PREFLIGHT_TIMEOUT", "90")) if timeout_s is None else timeout_s
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        ctl = control_group(timeout_s + 30.0)
        errs = []
        th = threading.Thread(target=self._preflight_body, args=(dev, errs), daemon=True, name="svd-preflight")
        th.start()
        th.join(timeout_s)
        hung = th.is_alive()
        if hung:
            errs = [f"timed out after {timeout_s:.0f} s inside a collective of the plan (backend {dist.get_backend()} hangs)"] + list(errs)
        err = "; ".join(errs) if errs else None
        flag = torch.tensor([0.0 if err is None else 1.0, 1.0 if hung else 0.0])
        try:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=ctl)
            failed, any_hung = flag[0].item() > 0, flag[1].item() > 0
        except Exception as e:                      # noqa: BLE001 -- a rank that never arrives (control-plane timeout)
            failed, any_hung, err = True, True, err or f"control plane: {type(e).__name__}: {e}"
        if any_hung:
            _CONTROL["use"] = True
            self.abort_report = self._abort_data_path()
        if not failed:
            return None
        return "job-plan collective preflight failed on " + (f"this rank: {err}" if err else "another rank")

    def _data_path_groups(self):
        gs = [dist.group.WORLD]
        for holder in (self.cfg_exchange, self.sp):
            g = getattr(holder, "group", None)
            if g is not None and g not in gs:
                gs.append(g)
        if self.decode_group is not None and self.decode_group not in gs:
            gs.append(self.decode_group)
        return gs

    def _abort_data_path(self, timeout_s=20.0):
        """After a collective of the plan hung.  What is recoverable: a HOST-side hang (bootstrap, a rank that never arrives: the helper thread stays parked, nothing
        runs on the device) needs nothing more -- the replicas run on, barrier and max-over-ranks on the gloo control group.  A DEVICE-side hang (a collective kernel
        spinning on a peer) would block the bench's device-wide torch.cuda.synchronize() for good: the RCCL communicators of the data-path groups are aborted here
        (ProcessGroupNCCL.abort = ncclCommAbort, which ends such a kernel), on a helper thread joined with a timeout because the abort itself can block.  Returns a
        short report; if the abort does not come back the bench line may not be produced (the backend's watchdog ends the process) -- INTEGRATION.md says so."""
        import threading
        if not torch.cuda.is_available():
            return "no device: host-side hang, nothing to abort"
        done = []

        def body():
            for g in self._data_path_groups():
                try:
                    if dist.get_backend(g) != "nccl":
                        continue
                    g._get_backend(torch.device("cuda")).abort()
                    done.append("aborted")
                except Exception as e:              # noqa: BLE001 -- reported, never raised: the replicas must still run
                    done.append(f"{type(e).__name__}: {e}")
        th = threading.Thread(target=body, daemon=True, name="svd-abort")
        th.start()
        th.join(timeout_s)
        return ("abort timed out; " if th.is_alive() else "") + (", ".join(done) if done else "no RCCL communicator")

    def _preflight_body(self, dev, errs):
        # Every rank issues EVERY collective of the sequence, in the same order, whatever its local checks say: a rank that stopped at a failed
        # check would leave the others waiting inside the next collective (a mismatched sequence hangs RCCL instead of raising).  Local
        # failures -- exceptions of the collectives AND of the glue around them (tensor construction, slicing, the final read-back, which can
        # surface an asynchronous device error) -- are only RECORDED here and exchanged by the flag all-reduce of _preflight.

        def step(what, fn, fallback=None):
            try:
                return fn()
            except Exception as e:                  # noqa: BLE001 -- whatever the backend throws, the job must still produce a number
                errs.append(f"{what}: {type(e).__name__}: {e}")
                return fallback() if fallback is not None else None

        def expect(cond, what):
            ok = step(what, cond)
            if ok is not None and not ok:
                errs.append(what)

        if dev.type == "cuda":
            step("set_device", lambda: torch.cuda.set_device(dev))              # the current device is thread-local
        net = step("tensor construction", lambda: torch.full((4, 4), float(self.rank), device=dev), lambda: torch.zeros((4, 4)))
        got = step("cfg all-gather", lambda: self.cfg_exchange.gather(net))
        expect(lambda: got is not None and got.shape == (8, 4), "cfg all-gather shape")
        if self.sp is not None:
            sp, T, pix, C = self.sp, 2 * self.sp.size + 1, 2 * self.sp.size, 8
            mk = lambda d: torch.arange(T * pix * C, dtype=torch.float32, device=d).reshape(T * pix, C).to(torch.float16)
            full = step("tensor construction", lambda: mk(dev), lambda: mk("cpu"))
            mine = step("take_frames", lambda: sp.take_frames(full, 1, T, pix), lambda: sp.take_frames(mk("cpu"), 1, T, pix))
            px = step("all-to-all (frames -> pixels)", lambda: sp.to_pixels(mine, 1, T, pix))
            if px is None:                              # keep the sequence aligned: the inverse exchange still runs, on a stand-in of the right shape
                px = torch.zeros((T * sp.pix_local(pix), C), dtype=mine.dtype, device=mine.device)
            back = step("all-to-all (pixels -> frames)", lambda: sp.to_frames(px, 1, T, pix))
            expect(lambda: back is not None and torch.equal(back, mine), "all-to-all round trip")
            gf = step("padded all-gather", lambda: sp.gather_frames(mine, 1, T, pix))
            expect(lambda: gf is not None and torch.equal(gf, full), "padded all-gather")
            ones = step("tensor construction", lambda: torch.ones((1, 32, 2), dtype=torch.float64, device=dev), lambda: torch.ones((1, 32, 2), dtype=torch.float64))
            sums = step("fp64 all-reduce", lambda: sp.allreduce_sums(ones))
            expect(lambda: sums is not None and float(sums[0, 0, 0]) == sp.size, "fp64 all-reduce")
        t = step("tensor construction", lambda: torch.full((3, 5), float(self.rank), device=dev), lambda: torch.full((3, 5), float(self.rank)))
        src = step("get_global_rank", lambda: dist.get_global_rank(self.decode_group, 0), lambda: 0)
        step("broadcast", lambda: broadcast(t, src=src, group=self.decode_group))
        if dev.type == "cuda":
            step("synchronize", torch.cuda.synchronize)
        expect(lambda: float(t[0, 0]) == float(src), "broadcast")

    @classmethod
    def from_env(cls, world, mode, **kw):
        rank = dist.get_rank() if (dist.is_initialized() and world > 1) else 0
        return cls(world, rank, mode, **kw)

    @property
    def scaling(self):
        # replicas: work grows with the GPUs; CFG pairs: one video on 2 GPUs is strong scaling, more pairs add videos (weak)
        if self.world == 1:
            return "strong"
        if self.mode == "replica" or (self.mode == "pairs" and self.world > 2):
            return "weak"
        return "strong"

    def attach(self, wrapper, vae):
        """Give the networks their share of the plan: the StreamingWrapper runs its forward sequence-parallel over `sp`; the decoder
        shards its frame groups over `decode_group`."""
        wrapper.sp = self.sp
        if hasattr(vae, "decode_group"):
            vae.decode_group = self.decode_group if self.mode != "replica" else None

    def describe(self):
        if self.world == 1:
            return "single GPU"
        if self.mode == "replica":
            why = f"; fell back from the one-job plan: {self.fallback_reason}" if self.fallback_reason else ""
            if self.abort_report:
                why += f" [data-path communicators: {self.abort_report}]"
            return f"replica-per-gpu x{self.world} (independent videos, no data-path collective{why})"
        if self.mode == "pairs":
            return (f"{self.n_videos} independent video(s), each on a CFG pair of GPUs (ranks 2k | 2k+1 evaluate the unconditional | conditional half; "
                    f"one RCCL all-gather of the 3.7 MB network output per Euler step; decode frame groups split inside the pair); no collective "
                    f"between pairs")
        sp = self.sp.size if self.sp else 1
        return (f"one job over {self.world} GPUs: CFG pair (RCCL all-gather of the network output per Euler step) x frame<->pixel sequence "
                f"parallelism of degree {sp} (RCCL all-to-all around the temporal operators, all-reduce of the 5-D GroupNorm sums), "
                f"decode frame groups sharded over all ranks")
