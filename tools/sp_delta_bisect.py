#!/usr/bin/env python
"""Where does the sequence-parallel forward start to differ from the single-process forward?  (round-3 review, weak #2)

    python tools/sp_delta_bisect.py [--world 2] > gpurun_out/sp_delta_bisect.txt        # on an MI355X; the ranks share cuda:0 over gloo

Per-operator bisect of tests/test_gpu_multiproc.py's case (tiny StreamingWrapper, T = 7 | Tc = 3, uneven frame ranges) on the REAL HIP
kernels.  Every VideoResBlock / SpatialVideoTransformer / CAM-merger output of three forwards is captured in execution order:
  A  single process;
  B  sequence parallel (degree 2), each module's local frames all-gathered back into the single-process row order;
  C  single process again, but with the (mean, rstd) of every POOLED GroupNorm (time_stack norms, CAM norm: the only statistics whose
     summation order differs under sequence parallelism) nudged by one fp32 ulp -- no other change.
and printed as relative L2 of B vs A and C vs A per module, next to the final outputs' error against the fp32 CPU oracle.
If B's first non-zero row is a module whose only sharding-dependent input is such a statistic and C grows the same way, the delta is the
amplification of 16-bit rounding flips downstream of a 1e-7 perturbation, not a different computation."""
import argparse
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
T, TC, H, W = 7, 3, 16, 16


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _inputs():
    g = torch.Generator(); g.manual_seed(77)
    r = lambda *s: torch.randn(*s, generator=g)
    F = 2 * T
    return dict(x=r(F, 4, H, W), t=r(F) * 0.5, concat=r(F, 4, H, W) * 0.5, crossattn=r(2, 1, 1024).repeat_interleave(T, 0),
                vector=r(2, 768).repeat_interleave(T, 0) * 0.5, ctrl_frames=torch.rand(1, TC, 3, 8 * H, 8 * W, generator=g) * 2 - 1)


def _rel(a, b):
    d = b.float().pow(2).mean().sqrt().item()
    return ((a.float() - b.float()).pow(2).mean().sqrt().item() / d) if d > 0 else 0.0


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    torch.cuda.set_device(0)
    torch.set_grad_enabled(False)
    from oracle import cases
    from streamingt2v_amd import lib as L, ops, parallel, video_model as VM
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.wrappers import StreamingWrapper
    assert parallel.init_from_env(backend="gloo") == world
    try:
        tu = cases.TINY_UNET
        cfg = VM.UNetConfig(num_res_blocks=tu["num_res_blocks"], attention_resolutions=tu["attention_resolutions"], channel_mult=tu["channel_mult"],
                            conditioning_embedding_out_channels=tu["cond_embed"])
        unet, cn = VM.VideoUNet(cfg), VM.ControlNet(cfg)
        sd_u, sd_c = init_by_name(unet.spec(), seed=1), init_by_name(cn.spec(), seed=2)
        unet.load_state_dict(sd_u, device="cuda")
        cn.load_state_dict(sd_c, device="cuda")
        win = {k: v.cuda() for k, v in _inputs().items()}
        c = {k: win[k] for k in ("concat", "crossattn", "vector")}
        kw = dict(batch_size=2, num_video_frames=T, image_only_indicator=torch.zeros(2, T, device="cuda"), ctrl_frames=win["ctrl_frames"])
        wrap = StreamingWrapper(unet, cn, TC)

        # ---- capture hooks ----
        log = []

        def hook(cls, kind):
            orig = cls.forward

            def fwd(self, x, *a, **k):
                y = orig(self, x, *a, **k)
                sp = k.get("sp")
                if kind == "cam":           # forward(sample, cond, F, T, Tc, H, W, sp=)
                    F_, T_, Hh, Ww = a[1], a[2], a[4], a[5]
                else:                       # forward(x, emb|ctx, F|tctx, T, H, W, sp=) : ResBlock (x, emb, F, T, H, W) ; SVT (x, ctx, tctx, F, T, H, W)
                    F_, T_, Hh, Ww = (a[1], a[2], a[3], a[4]) if kind == "res" else (a[2], a[3], a[4], a[5])
                full = y
                if sp is not None:
                    tl = sp.frame_counts(T_)[sp.rank]
                    full = sp.gather_frames(y, F_ // tl, T_, Hh * Ww)
                log.append((self.p + f" [{kind} T={T_} {Hh}x{Ww}]", full.float().clone()))
                return y
            cls.forward = fwd
            return orig
        origs = [(VM.VideoResBlock, hook(VM.VideoResBlock, "res")), (VM.SpatialVideoTransformer, hook(VM.SpatialVideoTransformer, "svt")),
                 (VM.ConditionalModel, hook(VM.ConditionalModel, "cam"))]

        def run(sp=None):
            log.clear()
            wrap.sp = sp
            wrap.reset_caches()
            y = wrap.forward(win["x"], win["t"], c, **kw)
            return y.float().clone(), list(log)

        yA, A = run()
        yA2, A2 = run()
        sp_groups = [dist.new_group([2 * k, 2 * k + 1]) for k in range(world // 2)]
        yB, B = run(parallel.SeqParallel(sp_groups[rank // 2]))
        # ---- C: one-ulp nudge of the pooled statistics, single process ----
        gn0 = ops.groupnorm

        def gn_nudged(x, frames, pix, gamma, beta, eps, *, frames_per_stat=1, silu=False, groups=32, out=None):
            if frames_per_stat == 1:
                return gn0(x, frames, pix, gamma, beta, eps, frames_per_stat=frames_per_stat, silu=silu, groups=groups, out=out)
            rows, ld = x.shape[0], x.stride(0)
            Cc = x.shape[1]
            partial, stats = ops._gn_workspace(x.device, frames, Cc, frames // frames_per_stat, groups)
            L.check(L.lib.svd_groupnorm_stats(ops._p(x), ld, frames, pix, Cc, groups, frames_per_stat, float(eps), ops._p(partial), ops._p(stats),
                                              ops._dt_in(x), ops._stream()), "svd_groupnorm_stats")
            n = 2 * groups * (frames // frames_per_stat)
            stats[:n].mul_(1.0 + 2.0 ** -23)              # TEST PERTURBATION: (mean, rstd) one fp32 ulp up
            if out is None:
                out = torch.empty((rows, Cc), dtype=ops._odt(x), device=x.device)
            L.check(L.lib.svd_groupnorm_apply(ops._p(x), ld, ops._p(out), out.stride(0), frames, pix, Cc, groups, frames_per_stat, ops._p(stats),
                                              ops._p(gamma), ops._p(beta), int(silu), ops._dt_in(x), ops._stream()), "svd_groupnorm_apply")
            return out
        ops.groupnorm = gn_nudged
        yC, Cl = run()
        ops.groupnorm = gn0
        for cls, o in origs:
            cls.forward = o
        res = None
        if rank == 0:
            from oracle import svd_oracle as O
            ocfg = O.Cfg(num_res_blocks=tu["num_res_blocks"], attention_resolutions=tu["attention_resolutions"], channel_mult=tu["channel_mult"],
                         cond_embed_channels=tu["cond_embed"])
            cpu = {k: v.cpu() for k, v in win.items()}
            yo = O.streaming_wrapper(sd_u, sd_c, ocfg, cpu["x"], cpu["t"], {k: cpu[k] for k in ("concat", "crossattn", "vector")}, 2, T, TC,
                                     cpu["ctrl_frames"]).cuda()
            rows = []
            assert len(A) == len(B) == len(Cl) == len(A2)
            for (n, a), (_, a2), (_, b), (_, cc) in zip(A, A2, B, Cl):
                rows.append((n, _rel(a2, a), _rel(b, a), _rel(cc, a)))
            res = dict(rows=rows, final=dict(rerun=_rel(yA2, yA), sp=_rel(yB, yA), nudged=_rel(yC, yA), single_vs_oracle=_rel(yA, yo),
                                             sp_vs_oracle=_rel(yB, yo), nudged_vs_oracle=_rel(yC, yo)))
        torch.cuda.synchronize()
        out.put((rank, res))
    finally:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2)
    a = ap.parse_args()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, a.world, port, q)) for r in range(a.world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    r = res[0]
    print(f"# tools/sp_delta_bisect.py, world {a.world}: tiny StreamingWrapper (T = {T}, Tc = {TC}, {H}x{W} latent), fp16, HIP kernels; relative L2 against forward A")
    print(f"{'module output (execution order: ControlNet, then UNet)':72s} {'A rerun':>9s} {'B: SP':>9s} {'C: 1-ulp':>9s}")
    for n, e0, eb, ec in r["rows"]:
        print(f"{n:72s} {e0:9.2e} {eb:9.2e} {ec:9.2e}")
    f = r["final"]
    print(f"{'network output':72s} {f['rerun']:9.2e} {f['sp']:9.2e} {f['nudged']:9.2e}")
    print(f"network output vs the fp32 CPU oracle: single process {f['single_vs_oracle']:.3e}, sequence parallel {f['sp_vs_oracle']:.3e}, "
          f"1-ulp-nudged statistics {f['nudged_vs_oracle']:.3e}")


if __name__ == "__main__":
    main()
