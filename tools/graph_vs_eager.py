"""Is the captured step (TimeT.enable_step_graph) the eager step?  

  twin      two models A and B from the same weights train ``--steps`` steps on the same clips, each eagerly or through the graph
            (``--graph_a / --graph_b``), each with or without a host synchronisation between steps (``--sync_a / --sync_b``; without one
            the host runs ahead of the device - with graph replays by many steps: the bench's regime); losses, every gradient and every
            parameter must be bit for bit equal.
  localize  ONE model; at every step the eager forward + backward and the graph's run from the same weights, and the step's named
            intermediates (TimeT._debug_tensors) are compared in launch order: the first one that differs names the kernel.

    python tools/graph_vs_eager.py --config c2 --mode twin --steps 6 --sync_b 0
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CONFIGS = {   # arch, clips, frames, prototypes, teacher + queue rows
    "c1": ("dino-s16", 2, 2, 50, 0),
    "c2": ("dino-s16", 32, 4, 200, 0),
    "c3": ("dino-s16", 32, 4, 200, 2048),
    "c4": ("dino-b16", 16, 8, 400, 0),
    "c5": ("dino-s8", 16, 4, 200, 0),
}


def make(cfg, total_steps, prefill_queue=True):
    from timetuning_amd import synth
    from timetuning_amd.models import FeatureExtractor
    from timetuning_amd.my_utils import cosine_scheduler
    from timetuning_amd.time_tuning import SwavOptimizer, TimeT

    arch, bs, fs, K, queue = CONFIGS[cfg]
    fe = FeatureExtractor(arch, "", [1024, 1024, 512, 256], unfreeze_layers=["blocks.11", "blocks.10"], init="dino", return_attention=False)
    m = TimeT(fe, K, prototype_init=torch.from_numpy(synth.make_prototypes(K, fe.feature_dim))).cuda()
    o = SwavOptimizer(m, "AdamW", True, 1e-5, 1e-4, "CosineAnnealingLR", cosine_scheduler(0.04, 0.4, 1, total_steps), total_steps, 1)
    if queue:
        m.init_momentum_teacher()
        m.set_momentum_teacher_schedular_params(0.995, 1.0, 1, total_steps)
        m.init_queue(queue)
        if prefill_queue:
            g = torch.Generator().manual_seed(5)
            m.set_queue(torch.nn.functional.normalize(torch.randn(queue, fe.feature_dim, generator=g), dim=1).cuda())
    return m, o


def twin(a):
    """Models A and B from the same weights on the same clips; A = (graph_a, sync_a), B = (graph_b, sync_b)."""
    from timetuning_amd import synth

    arch, bs, fs, K, queue = CONFIGS[a.config]
    clips = [torch.from_numpy(synth.make_clips(bs, fs, 224, seed=40 + i)).cuda() for i in range(2 if a.same_clip else a.steps)]
    runs = []
    for graph, sync in ((a.graph_a, a.sync_a), (a.graph_b, a.sync_b)):
        m, o = make(a.config, a.total_steps or a.steps + 2)
        if graph:
            m.enable_step_graph()
        torch.manual_seed(123)
        rec = []
        static = None
        for i in range(a.steps):
            x = clips[0] if a.same_clip else clips[i]
            if a.keep:
                m._debug_tensors = dk = {}
            loss = m.get_loss(x)
            inter = None
            if a.keep:
                if dk:
                    static = dk    # (a captured step's tensors are every replay's)
                inter = {k: v.clone() for k, v in static.items()}
            m.train_update(o, loss, min(i + 1, a.steps) if queue else 0)
            rec.append(dict(loss=loss.detach().clone(), grads={n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None},
                            params={n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad}, inter=inter))
            if sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        runs.append((m, rec))
    (ma, ra), (mb, rb) = runs
    print(f"[twin {a.config}] A: graph={a.graph_a} sync={a.sync_a} | B: graph={a.graph_b} sync={a.sync_b}; graphs captured: {len(getattr(ma, '_step_graphs', {}))} / {len(getattr(mb, '_step_graphs', {}))}", flush=True)
    bad = 0
    for i, (A, B) in enumerate(zip(ra, rb)):
        gd = [n for n in A["grads"] if not torch.equal(A["grads"][n], B["grads"][n])]
        pd = [n for n in A["params"] if not torch.equal(A["params"][n], B["params"][n])]
        line = f"  step {i}: loss A {A['loss'].item():.6f} B {B['loss'].item():.6f} {'==' if A['loss'].item() == B['loss'].item() else '!='}"
        if A["inter"] is not None and B["inter"] is not None:
            idf = [k for k in A["inter"] if k in B["inter"] and A["inter"][k].shape == B["inter"][k].shape and not torch.equal(A["inter"][k], B["inter"][k])]
            line += f"; intermediates differing {len(idf)} / {len(A['inter'])}" + (f" ({', '.join(idf[:6])})" if idf else "")
        line += f"; gradients differing {len(gd)} / {len(A['grads'])}" + (f" ({', '.join(gd[:4])})" if gd else "")
        line += f"; parameters after the update differing {len(pd)} / {len(A['params'])}" + (f" ({', '.join(pd[:4])})" if pd else "")
        print(line, flush=True)
        bad += bool(gd) or bool(pd) or A["loss"].item() != B["loss"].item()
    return bad


def localize(a):
    from timetuning_amd import synth

    arch, bs, fs, K, queue = CONFIGS[a.config]
    x = torch.from_numpy(synth.make_clips(bs, fs, 224, seed=41)).cuda()
    m, o = make(a.config, a.steps + 2)
    m.enable_step_graph()
    perm = torch.randperm(bs * m.feature_extractor.spatial_resolution ** 2)
    bad = 0
    graph_keep = None
    for i in range(a.steps):
        qsave = (m.queue.clone(), m._queue_rows_pushed, m._queue_seen) if m.queue is not None else None
        # eager, from these weights
        m._step_graph_on = False
        m._debug_tensors = ek = {}
        m.zero_grad(set_to_none=True)
        le = m.get_loss(x, queue_perm=perm if queue else None)
        le.backward()
        torch.cuda.synchronize()
        ek = {k: v.clone() for k, v in ek.items()}
        ge = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        if qsave is not None:
            m.queue.copy_(qsave[0]); m._queue_rows_pushed = qsave[1]; m._queue_seen = m._queue_signature() if qsave[2] is not None else None
        # the graph, from the same weights
        m._step_graph_on = True
        m._debug_tensors = gk = {}
        m.zero_grad(set_to_none=True)
        lg = m.get_loss(x, queue_perm=perm if queue else None)
        torch.cuda.synchronize()
        if gk:
            graph_keep = gk          # the capture's tensors are the replay's
        m._debug_tensors = None
        inter = {k: v.clone() for k, v in graph_keep.items()} if graph_keep is not None else None   # (the backward overwrites some in place - as the eager one did)
        m.train_update(o, lg, min(i + 1, a.steps) if queue else 0)   # backward (hands out the graph's gradient buffers) + update
        gg = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        line = f"  step {i}: loss eager {le.item():.6f} graph {lg.item():.6f}"
        if inter is not None and len(m._step_graphs):
            first = None
            nd = 0
            for k in ek:
                if k in inter and ek[k].shape == inter[k].shape and not torch.equal(ek[k], inter[k]):
                    nd += 1
                    first = first or k
            line += f"; intermediates differing {nd} / {len(ek)}" + (f" first: {first}" if first else "")
            if first:
                d = (ek[first].double() - inter[first].double()).abs()
                line += f" (max abs {d.max().item():.3e}, {int((d > 0).sum())} of {d.numel()} elements)"
        gd = [n for n in ge if not torch.equal(ge[n], gg[n])]
        line += f"; gradients differing {len(gd)} / {len(ge)}" + (f" first: {gd[0]}" if gd else "")
        print(line, flush=True)
        bad += bool(gd) or le.item() != lg.item()
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--mode", default="twin", choices=["twin", "localize"])
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--graph_a", type=int, default=0)
    ap.add_argument("--sync_a", type=int, default=1)
    ap.add_argument("--graph_b", type=int, default=1)
    ap.add_argument("--sync_b", type=int, default=1)
    ap.add_argument("--total_steps", type=int, default=0, help="length of the lr / wd schedules (bench.py: steps + warmup + 200)")
    ap.add_argument("--keep", type=int, default=0, help="twin: also compare the step's named intermediates")
    ap.add_argument("--same_clip", type=int, default=1, help="the bench's regime: the same batch every step")
    ap.add_argument("--precision", default="f16x3")
    ap.add_argument("--in_flight", type=int, default=None, help="time_tuning.STEP_GRAPHS_IN_FLIGHT (0 = unbounded: reproduces the round-5 divergence)")
    a = ap.parse_args()
    from timetuning_amd import hip_ops, time_tuning

    if a.in_flight is not None:
        time_tuning.STEP_GRAPHS_IN_FLIGHT = a.in_flight

    hip_ops.set_gemm_precision(a.precision)
    bad = twin(a) if a.mode == "twin" else localize(a)
    print(f"[{a.mode} {a.config} {a.precision}] {'EQUAL' if not bad else 'DIFFERENT'}", flush=True)


if __name__ == "__main__":
    main()
