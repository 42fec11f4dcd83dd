#!/usr/bin/env python3
"""bench.py -- IAF-step throughput of the MI355X engine on BASELINE.json's configs[1].

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (config.workload): "cifar10 n_z=32 n_h=160 depths=[10,10] depth_ar=2 down_iaf2_nl bs=32":
one STEP = one pass of the IAF hot path (weight prep + ar_multiconv2d + affine transform +
log-det term, tf_train.py:69-72) over one synthetic batch of 32 through the model's layer schedule:
10 IAF layers on [32,32,16,16] latents and 10 on [32,32,8,8], each layer with its own weights,
z and context (seeded N(0,1) data, random-init weights; inputs resident in HBM).
value = samples/s through the whole IAF stack = N_gpus * 32 / t_step  (weak scaling: each rank
runs its own batch of 32; the forward path has no collective).

The timed region replays a hipGraph of the step (1 batched weight prep + 60 conv launches); right after it
the dominant kernel (the 160->160 masked conv at 16x16) is timed with HIP events on the same stream
(50 back-to-back launches per event pair, and per-launch brackets inside eager steps) for the roofline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (BASELINE config 2: 32)")
    ap.add_argument("--n-z", type=int, default=32)
    ap.add_argument("--n-h", type=int, default=160)
    ap.add_argument("--depth-ar", type=int, default=2)
    ap.add_argument("--depths", type=str, default="10,10")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--cached-weights", action="store_true",
                    help="prepare weights once outside the timed region (inference with frozen weights); "
                         "default re-derives mask*V / weight-norm every step like the reference graph does")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tune", type=str, default="", help="layer:nt,pxt,wco,ks;... launch-shape override")
    ap.add_argument("--layers", action="store_true",
                    help="widened workload (SURVEY 8f-4): whole IAFLayers (plain convs + posterior block), forward")
    ap.add_argument("--train", action="store_true",
                    help="extra mode (not the headline metric): data-parallel TRAINING step of the IAF posterior stack -- "
                         "posterior block forward + backward for every layer, one RCCL all-reduce of the flat gradient "
                         "buffer, fused Adamax + EMA")
    return ap.parse_args()


def make_layer_inputs(rng, B, n_z, n_h, depth_ar, H):
    import golden_inputs as gi
    params = gi.ar_multiconv2d_params(rng, n_z, [n_h] * depth_ar, [n_z, n_z])
    # SURVEY 8d synthetic inputs: V~N(0,.05^2), g,b~N(0,.1^2), z,context~N(0,1)
    for k in params:
        if k.endswith("/g") or k.endswith("/b"):
            params[k] = 0.1 * rng.standard_normal(params[k].shape)
    z = rng.standard_normal((B, n_z, H, H))
    ctx = rng.standard_normal((B, n_h, H, H))
    return params, z, ctx


def cpu_baseline(args, depths):
    """torch-CPU fp32 port (oracle/iaf_cpu_port.py) on a bounded sample of the same workload."""
    from oracle import iaf_cpu_port as Pt
    ncpu = os.cpu_count() or 1
    # oneDNN convs of this size stop scaling (and collapse when oversubscribed) well before a 128+-core host is
    # full: probe a few thread counts on the 16x16 step and keep the fastest -- that is the baseline we report.
    prng = np.random.RandomState(7)
    params, z, ctx = make_layer_inputs(prng, args.batch, args.n_z, args.n_h, args.depth_ar, 16)
    tp = Pt.as_torch(params)
    w = Pt.prepare_weights(tp, args.n_z, [args.n_h] * args.depth_ar)
    zt, ct = torch.from_numpy(z.astype(np.float32)), torch.from_numpy(ctx.astype(np.float32))
    best = None
    for nt in sorted(set(min(c, ncpu) for c in (4, 8, 16, 32, 64, 128))):
        torch.set_num_threads(nt)
        with torch.no_grad():
            Pt.iaf_step(zt, ct, w, args.depth_ar)
            t0 = time.perf_counter()
            for _ in range(3):
                Pt.iaf_step(zt, ct, w, args.depth_ar)
            dt = (time.perf_counter() - t0) / 3
        if best is None or dt < best[0]:
            best = (dt, nt)
        if dt > 4 * best[0]:
            break
    nthreads = best[1]
    torch.set_num_threads(nthreads)
    rng = np.random.RandomState(1)
    per_level = []
    budget = args.cpu_seconds / max(len(depths), 1)
    for lvl, _ in enumerate(depths):
        H = 16 >> lvl
        params, z, ctx = make_layer_inputs(rng, args.batch, args.n_z, args.n_h, args.depth_ar, H)
        tp = Pt.as_torch(params)
        zt, ct = torch.from_numpy(z.astype(np.float32)), torch.from_numpy(ctx.astype(np.float32))

        def one():
            w = tp_w if args.cached_weights else Pt.prepare_weights(tp, args.n_z, [args.n_h] * args.depth_ar)
            return Pt.iaf_step(zt, ct, w, args.depth_ar)

        tp_w = Pt.prepare_weights(tp, args.n_z, [args.n_h] * args.depth_ar)
        with torch.no_grad():
            for _ in range(3):
                one()
            ts = []
            t_end = time.perf_counter() + budget
            while time.perf_counter() < t_end or len(ts) < 5:
                t0 = time.perf_counter()
                one()
                ts.append(time.perf_counter() - t0)
        per_level.append(float(np.median(ts)))
    t_model = sum(d * t for d, t in zip(depths, per_level))
    return {
        "value": args.batch / t_model, "unit": "samples/s", "cores": nthreads, "kind": "port",
        "sample": "torch-CPU fp32 port of the same IAF step (oracle/iaf_cpu_port.py), B=%d: median of repeated single "
                  "IAF steps per level (%s ms at %s), ~%.0f s total, scaled to the depths=%s schedule; the reference has "
                  "no CPU path (SURVEY D1); %d of %d host CPUs used (fastest of a thread-count probe)" % (args.batch, ",".join("%.2f" % (1e3 * t) for t in per_level),
                                              ",".join("%dx%d" % (16 >> i, 16 >> i) for i in range(len(depths))),
                                              args.cpu_seconds, depths, nthreads, ncpu),
        "ms_per_iaf_step": [1e3 * t for t in per_level],
    }


def train_bench(args, depths, dist, rank, n_gpus):
    """DP training step of the IAF posterior stack (SURVEY 8f-1,2): per step, for every layer, posterior block forward
    (tf_train.py:56-85) + backward (what opt.compute_gradients derives, tf_train.py:138), gradients written into ONE flat
    buffer, all-reduce(sum) over ranks (RCCL), fused Adamax(1/N)+EMA (tf_utils/common.py:86, adamax.py:40-56,
    tf_train.py:157-158).  Synthetic upstream gradients (dz ~ N(0,1), dkl_obj = 1)."""
    import iaf_amd
    from iaf_amd import parallel as par
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    rng = np.random.RandomState(4321 + rank)
    wrng = np.random.RandomState(99)                      # identical initial weights on every rank
    layers, named = [], {}
    for lvl, nlayer in enumerate(depths):
        H = 16 >> lvl
        for j in range(nlayer):
            params, _, _ = make_layer_inputs(wrng, args.batch, args.n_z, args.n_h, args.depth_ar, H)
            f = lambda c: dev(rng.standard_normal((args.batch, c, H, H)))
            inp = dict(qm=f(args.n_z), ql=0.1 * f(args.n_z), rm=f(args.n_z), rl=0.1 * f(args.n_z), pm=f(args.n_z),
                       pl=0.1 * f(args.n_z), uc=f(args.n_h), dc=f(args.n_h), eps=f(args.n_z), dz=f(args.n_z),
                       dko=torch.ones(args.batch, device="cuda"))
            st = iaf_amd.ARStack(args.n_z, [args.n_h] * args.depth_ar)
            st.set_training(True)
            pre = "IAF_%d_%d/ar_multiconv2d/" % (lvl, j)
            for k, v in params.items():
                named[pre + k] = dev(v)
            layers.append(dict(stack=st, pre=pre, keys=list(params), inp=inp))
    flat = par.FlatParams(named)
    for L in layers:
        L["params"] = {k: flat.p[L["pre"] + k] for k in L["keys"]}
        L["gradviews"] = {k: flat.g[L["pre"] + k] for k in L["keys"]}
    prep = iaf_amd.PrepBatch([L["stack"] for L in layers])
    wnb = iaf_amd.WnBwdBatch(stacks=[L["stack"] for L in layers])     # mask + weight-norm backward: one launch per model
    plist = [L["params"] for L in layers]
    glist = [L["gradviews"] for L in layers]

    def step():
        compute()
        flat.all_reduce_grads()
        flat.adamax_ema_step(1e-4, world=n_gpus)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # the compute part of the step (everything but the gradient exchange and the update) replays as one hipGraph; the
    # all-reduce and the fused Adamax+EMA follow on the same stream
    def compute():
        prep.run(plist)
        for L in layers:
            i, st = L["inp"], L["stack"]
            fw = st.posterior_block_train(i["qm"], i["ql"], i["rm"], i["rl"], i["pm"], i["pl"], i["uc"], i["dc"], i["eps"], 0.25)
            st.posterior_block_backward(i["qm"], i["ql"], i["rm"], i["rl"], i["pm"], i["pl"], i["eps"], 0.25, fw["z"], i["dz"],
                                        i["dko"], L["params"], grads_out=L["gradviews"])
        wnb.run(stack_params=plist, stack_grads=glist)

    stream = torch.cuda.Stream()
    graph = None
    with torch.cuda.stream(stream):
        step()
        stream.synchronize()
        if not args.no_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                compute()

            def step():  # noqa: F811
                graph.replay()
                flat.all_reduce_grads()
                flat.adamax_ema_step(1e-4, world=n_gpus)
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "IAF posterior-stack TRAIN-step samples/sec (forward + backward + grad all-reduce + Adamax/EMA)",
            "value": n_gpus * args.batch / (elapsed / args.steps), "unit": "samples/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cifar10 n_z=%d n_h=%d depths=%s depth_ar=%d down_iaf2_nl bs=%d per GPU, kl_min=0.25; "
                                   "%d trainable fp32 parameters in one flat all-reduce bucket (%.1f MB)"
                                   % (args.n_z, args.n_h, depths, args.depth_ar, args.batch, flat.params.numel(),
                                      4e-6 * flat.params.numel()),
                       "global_batch": n_gpus * args.batch,
                       "launch": "hipGraph replay of forward+backward, then all-reduce + update" if graph is not None else "eager",
                       "parallelism": "dp%d (RCCL all-reduce of one flat gradient bucket)" % n_gpus}}))


def layers_bench(args, depths, dist, rank, n_gpus):
    """SURVEY 8f-4 widening: whole non-downsampling IAFLayers (tf_train.py:23-95), forward, mode "train": per step the
    bottom-up pass (`up`, chained within a level) then the top-down pass (`down`, chained within a level) of
    sum(depths) layers; all weight-norm reparametrisations re-derived every step in two batched launches."""
    import golden_inputs as gi
    import iaf_amd
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    rng = np.random.RandomState(777 + rank)
    wrng = np.random.RandomState(5)
    zs, hs, B = args.n_z, args.n_h, args.batch
    levels = []
    for lvl, nlayer in enumerate(depths):
        H = 16 >> lvl
        L = []
        for _ in range(nlayer):
            p = {}
            for nm, (ci, co) in (("up_conv1", (hs, 2 * zs + 2 * hs)), ("up_conv3", (hs, hs)),
                                 ("down_conv1", (hs, 4 * zs + 2 * hs)), ("down_conv2", (hs + zs, hs))):
                for k, v in gi.conv_params(wrng, ci, co).items():
                    p[nm + "/" + k] = dev(v)
            for k, v in gi.ar_multiconv2d_params(wrng, zs, [hs] * args.depth_ar, [zs, zs]).items():
                p["ar_multiconv2d/" + k] = dev(v)
            layer = iaf_amd.IAFLayer(zs, hs, depth_ar=args.depth_ar, kl_min=0.25)
            layer.load(p)
            L.append(dict(layer=layer, params=p, eps=dev(rng.standard_normal((B, zs, H, H)))))
        levels.append(dict(H=H, layers=L, up_in=dev(rng.standard_normal((B, hs, H, H))),
                           down_in=dev(rng.standard_normal((B, hs, H, H)))))
    all_layers = [L for lv in levels for L in lv["layers"]]
    prep_s = iaf_amd.PrepBatch([L["layer"].posterior.stack for L in all_layers])
    prep_c = iaf_amd.ConvPrepBatch([c for L in all_layers for c in L["layer"].convs()])
    splist = [iaf_amd.IAFLayer.stack_params(L["params"]) for L in all_layers]
    cplist = [t for L in all_layers for t in iaf_amd.IAFLayer.conv_params(L["params"])]

    def step(autotune=False):
        if not args.cached_weights:
            prep_s.run(splist)
            prep_c.run(cplist)
        outs = []
        for lv in levels:                         # bottom-up (tf_train.py:188-192)
            h = lv["up_in"]
            for L in lv["layers"]:
                h = L["layer"].up(h, autotune=autotune)
        for lv in reversed(levels):               # top-down (tf_train.py:195-200)
            h = lv["down_in"]
            for L in reversed(lv["layers"]):
                h, kl_obj, kl_cost = L["layer"].down(h, L["eps"], autotune=autotune)
                outs.append((kl_obj, kl_cost))
            outs.append(h)
        return outs

    stream = torch.cuda.Stream()
    graph = None
    with torch.cuda.stream(stream):
        step()
        step(autotune=True)                       # launch-shape search of the plain convs (cuDNN's algorithm search)
        stream.synchronize()
        if not args.no_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                keep = step()
        run = graph.replay if graph is not None else step

        def barrier():
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(args.warmup):
            run()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        barrier()
        elapsed = time.perf_counter() - t0

        # dominant kernel of this mode: down_conv1 (n_h -> 4 n_z + 2 n_h) at 16x16; 50 back-to-back launches per layer
        # between one event pair on the launch stream
        kt = []
        for L in levels[0]["layers"]:
            cv = L["layer"].down_conv1
            x = levels[0]["down_in"]
            call = lambda: cv(x, elu_input=True, split=[zs] * 4 + [hs] * 2)
            call()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(50):
                call()
            b.record(stream)
            b.synchronize()
            kt.append(a.elapsed_time(b) / 50)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return
    cv = levels[0]["layers"][0]["layer"].down_conv1
    fl, by = cv.work(B, 16, 16)
    k_ms = float(np.mean(kt))
    achieved = fl / (k_ms * 1e-3) / 1e12
    total_fl = 0.0
    for lv in levels:
        for L in lv["layers"]:
            total_fl += sum(c.work(B, lv["H"], lv["H"])[0] for c in L["layer"].convs())
            total_fl += L["layer"].posterior.stack.step_work(B, lv["H"], lv["H"])["live_flops"]
    print(json.dumps({
        "metric": "IAFLayer forward samples/sec (up + down of every layer: 4 plain weight-normed convs + IAF posterior block)",
        "value": n_gpus * B / (elapsed / args.steps), "unit": "samples/s", "n_gpus": n_gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "cifar10 z_size=%d h_size=%d depths=%s depth_ar=%d bs=%d per GPU, kl_min=0.25: %d non-downsampling "
                               "IAFLayers (tf_train.py:23-95), up pass then down pass, chained within a level"
                               % (zs, hs, depths, args.depth_ar, B, len(all_layers)),
                   "global_batch": n_gpus * B, "launch": "hipGraph replay" if graph is not None else "eager",
                   "weights": "re-derived every step (2 batched launches)" if not args.cached_weights else "prepared once",
                   "live_gflop_per_step": total_fl / 1e9,
                   "model_tflops": total_fl / (elapsed / args.steps) / 1e12,
                   "parallelism": "dp%d (batch-sharded replicas, no forward collective)" % n_gpus},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                     "kernel": "iaf_conv_kernel<.., EPI_PLAIN, 9 taps> (down_conv1 %d->%d, B=%d 16x16)" % (cv.n_in, cv.n_out, B),
                     "avg_launch_us": 1e3 * k_ms, "launches_timed": 50 * len(kt), "flops_per_launch": fl,
                     "bytes_per_launch": by, "hbm_frac_at_this_rate": (by / (k_ms * 1e-3) / 1e9) / PEAK_HBM_GBS,
                     "timing": "HIP events on the launch stream around 50 back-to-back launches per 16x16 layer"}}))


def layers_train_bench(args, depths, dist, rank, n_gpus):
    """DP training step of whole IAFLayers (SURVEY 8f-4 + 8f-1,2): weight prep, up pass, down pass, backward of both
    passes (every plain conv and the posterior block), gradients written into ONE flat buffer, all-reduce(sum) over
    ranks, fused Adamax(1/N)+EMA.  Synthetic upstream gradients (d output ~ N(0,1), d kl_obj = 1)."""
    import golden_inputs as gi
    import iaf_amd
    from iaf_amd import parallel as par
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    rng = np.random.RandomState(4321 + rank)
    wrng = np.random.RandomState(99)
    zs, hs, B = args.n_z, args.n_h, args.batch
    named, levels = {}, []
    for lvl, nlayer in enumerate(depths):
        H = 16 >> lvl
        L = []
        for j in range(nlayer):
            pre = "IAF_%d_%d/" % (lvl, j)
            for nm, (ci, co) in (("up_conv1", (hs, 2 * zs + 2 * hs)), ("up_conv3", (hs, hs)),
                                 ("down_conv1", (hs, 4 * zs + 2 * hs)), ("down_conv2", (hs + zs, hs))):
                for k, v in gi.conv_params(wrng, ci, co).items():
                    named[pre + nm + "/" + k] = dev(v)
            for k, v in gi.ar_multiconv2d_params(wrng, zs, [hs] * args.depth_ar, [zs, zs]).items():
                named[pre + "ar_multiconv2d/" + k] = dev(v)
            layer = iaf_amd.IAFLayer(zs, hs, depth_ar=args.depth_ar, kl_min=0.25)
            layer.set_training(True)
            L.append(dict(layer=layer, pre=pre, eps=dev(rng.standard_normal((B, zs, H, H)))))
        f = lambda: dev(rng.standard_normal((B, hs, H, H)))
        levels.append(dict(H=H, layers=L, up_in=f(), down_in=f(), d_up=f(), d_down=f()))
    flat = par.FlatParams(named)
    all_layers = [L for lv in levels for L in lv["layers"]]
    for L in all_layers:
        n = len(L["pre"])
        L["params"] = {k[n:]: v for k, v in flat.p.items() if k.startswith(L["pre"])}
        L["grads"] = {k[n:]: v for k, v in flat.g.items() if k.startswith(L["pre"])}
    prep_s = iaf_amd.PrepBatch([L["layer"].posterior.stack for L in all_layers])
    prep_c = iaf_amd.ConvPrepBatch([c for L in all_layers for c in L["layer"].convs()])
    splist = [iaf_amd.IAFLayer.stack_params(L["params"]) for L in all_layers]
    cplist = [t for L in all_layers for t in iaf_amd.IAFLayer.conv_params(L["params"])]
    dko = torch.ones(B, device="cuda")
    wnb = iaf_amd.WnBwdBatch(stacks=[L["layer"].posterior.stack for L in all_layers],
                             convs=[c for L in all_layers for c in L["layer"].convs()])
    sglist = [iaf_amd.IAFLayer.stack_params(L["grads"]) for L in all_layers]
    cglist = [t for L in all_layers for t in iaf_amd.IAFLayer.conv_params(L["grads"])]

    def compute(tune=False):
        prep_s.run(splist)
        prep_c.run(cplist)
        for lv in levels:
            h = lv["up_in"]
            for L in lv["layers"]:
                h = L["layer"].up_train(h, autotune=tune)
        for lv in reversed(levels):
            h = lv["down_in"]
            for L in reversed(lv["layers"]):
                h, _, _ = L["layer"].down_train(h, L["eps"], autotune=tune)
        for lv in levels:                                   # backward of the top-down pass, in reverse
            d = lv["d_down"]
            for L in lv["layers"]:
                d = L["layer"].down_backward(d, dko, L["params"], L["grads"], autotune=tune)
        for lv in reversed(levels):                         # backward of the bottom-up pass, in reverse
            d = lv["d_up"]
            for L in reversed(lv["layers"]):
                d = L["layer"].up_backward(d, L["params"], L["grads"], autotune=tune)
        wnb.run(stack_params=splist, stack_grads=sglist, conv_params=cplist, conv_grads=cglist)

    def step():
        compute()
        flat.all_reduce_grads()
        flat.adamax_ema_step(1e-4, world=n_gpus)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    stream = torch.cuda.Stream()
    graph = None
    with torch.cuda.stream(stream):
        step()
        compute(tune=True)             # launch-shape search of the plain convs, forward and data gradient (cuDNN's autotune)
        stream.synchronize()
        if not args.no_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                compute()

            def step():  # noqa: F811
                graph.replay()
                flat.all_reduce_grads()
                flat.adamax_ema_step(1e-4, world=n_gpus)
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        fwd_fl = 0.0
        for lv in levels:
            for L in lv["layers"]:
                fwd_fl += sum(c.work(B, lv["H"], lv["H"])[0] for c in L["layer"].convs())
                fwd_fl += L["layer"].posterior.stack.step_work(B, lv["H"], lv["H"])["live_flops"]
        print(json.dumps({
            "metric": "IAFLayer TRAIN-step samples/sec (forward + backward of every layer + grad all-reduce + Adamax/EMA)",
            "value": n_gpus * B / (elapsed / args.steps), "unit": "samples/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cifar10 z_size=%d h_size=%d depths=%s depth_ar=%d bs=%d per GPU, kl_min=0.25: %d "
                                   "non-downsampling IAFLayers; %d trainable fp32 parameters in one flat all-reduce bucket (%.1f MB)"
                                   % (zs, hs, depths, args.depth_ar, B, len(all_layers), flat.params.numel(),
                                      4e-6 * flat.params.numel()),
                       "global_batch": n_gpus * B,
                       "launch": "hipGraph replay of forward+backward, then all-reduce + update" if graph is not None else "eager",
                       "model_tflops_fwd_plus_bwd": 3.0 * fwd_fl / (elapsed / args.steps) / 1e12,
                       "parallelism": "dp%d (RCCL all-reduce of one flat gradient bucket)" % n_gpus}}))


def main():
    args = parse()
    depths = [int(d) for d in args.depths.split(",") if d]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("IAF_BENCH_FORCE_DIST"):      # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        # RCCL prints a banner (ROCm version / hostname / library path) to STDOUT when its first communicator comes up;
        # the contract is ONE JSON line on stdout, so that happens with fd 1 pointed at stderr
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    else:
        dist = None
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    n_gpus = world

    import iaf_amd
    iaf_amd._capi.lib()           # fail loudly if the HIP engine is not built
    if args.layers and args.train:
        layers_train_bench(args, depths, dist, rank, n_gpus)
        if dist is not None:
            dist.destroy_process_group()
        return
    if args.layers:
        layers_bench(args, depths, dist, rank, n_gpus)
        if dist is not None:
            dist.destroy_process_group()
        return
    if args.train:
        train_bench(args, depths, dist, rank, n_gpus)
        if dist is not None:
            dist.destroy_process_group()
        return
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()

    # ---------------- build the layer schedule (every rank its own data: seed + rank)
    rng = np.random.RandomState(1234 + rank)
    layers = []
    for lvl, nlayer in enumerate(depths):
        H = 16 >> lvl
        for _ in range(nlayer):
            params, z, ctx = make_layer_inputs(rng, args.batch, args.n_z, args.n_h, args.depth_ar, H)
            st = iaf_amd.ARStack(args.n_z, [args.n_h] * args.depth_ar)
            dp = {k: dev(v) for k, v in params.items()}
            zd, cd = dev(z), dev(ctx)
            out = (torch.empty_like(zd), torch.empty_like(zd))
            st.prepare(dp)
            layers.append(dict(stack=st, params=dp, z=zd, ctx=cd, out=out, H=H))
    if args.tune:
        for item in args.tune.split(";"):
            lay, shp = item.split(":")
            nt, pxt, wco, ks = [int(v) for v in shp.split(",")]
            for L in layers:
                L["stack"].set_tuning(int(lay), nt, pxt, wco, ks)

    prep = iaf_amd.PrepBatch([L["stack"] for L in layers])
    plist = [L["params"] for L in layers]

    def step():
        if not args.cached_weights:
            prep.run(plist)      # mask*V, l2-normalise, exp(g), repack (layers.py:56-60): all 20 layers, one launch
        for L in layers:
            L["stack"].iaf_step(L["z"], L["ctx"] if args.depth_ar > 0 else None, out=L["out"])

    stream = torch.cuda.Stream()
    graph = None
    with torch.cuda.stream(stream):
        step()                                                   # allocate workspaces, warm caches
        stream.synchronize()
        if not args.no_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                step()
        run = graph.replay if graph is not None else step

        def barrier():
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(args.warmup):
            run()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        barrier()
        elapsed = time.perf_counter() - t0

        # ---------------- kernel leg: eager steps, HIP events around every launch of the dominant kernel
        dom_layer = max(args.depth_ar - 1, 0)                    # the n_h -> n_h masked conv (layer 1 at depth_ar=2)
        prof = [L for L in layers if L["H"] == 16]
        ksteps = max(10, min(args.steps, 50))
        for L in prof:
            L["stack"].profile_enable(dom_layer, ksteps + 4)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        for L in prof:
            L["stack"].profile_read()
        for _ in range(ksteps):
            step()
        torch.cuda.synchronize()
        kms = []
        for L in prof:
            kms += L["stack"].profile_read()
            L["stack"].profile_enable(-1, 0)
        # primary figure: N back-to-back launches of the dominant kernel between ONE event pair (no per-launch
        # event/dispatch latency), averaged over the 16x16 layers
        kbatch = [L["stack"].time_layer(dom_layer, L["z"], L["ctx"] if args.depth_ar > 0 else None, reps=50) for L in prof]

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = 1e3 * elapsed / args.steps
    value = n_gpus * args.batch / (elapsed / args.steps)
    n_iaf = sum(depths)

    st0 = prof[0]["stack"]
    lw = st0.layer_work(dom_layer, args.batch, 16, 16)
    k_avg_ms = float(np.mean(kbatch))
    k_brk_ms = float(np.mean(kms)) if kms else float("nan")
    achieved = lw["live_flops"] / (k_avg_ms * 1e-3) / 1e12
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_dominant_kernel.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {
        "bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
        "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
        "kernel": "iaf_conv_kernel (masked 3x3 conv %d->%d, B=%d 16x16, GEMM layer %d)" % (args.n_h, args.n_h, args.batch, dom_layer),
        "avg_launch_us": 1e3 * k_avg_ms, "launches_timed": 50 * len(kbatch),
        "timing": "HIP events on the launch stream around 50 back-to-back launches per 16x16 layer (includes the "
                  "inter-launch gap); per-launch event brackets inside full steps read %.2f us over %d launches "
                  "(adds event/dispatch latency)" % (1e3 * k_brk_ms, len(kms)),
        "flops_per_launch_live": lw["live_flops"], "flops_per_launch_dense9tap": lw["dense_flops"],
        "bytes_per_launch": lw["bytes"],
        "hbm_frac_at_this_rate": (lw["bytes"] / (k_avg_ms * 1e-3) / 1e9) / PEAK_HBM_GBS,
    }
    work16 = st0.step_work(args.batch, 16, 16)
    out = {
        "metric": "IAF-step samples/sec (down_iaf2_nl posterior stack, forward + log-det)",
        "value": value, "unit": "samples/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "cifar10 n_z=%d n_h=%d depths=%s depth_ar=%d down_iaf2_nl bs=%d per GPU (BASELINE configs[1]); "
                        "one step = %d IAF steps: %s" % (args.n_z, args.n_h, depths, args.depth_ar, args.batch, n_iaf,
                                                        " + ".join("%d x [%d,%d,%d,%d]" % (d, args.batch, args.n_z, 16 >> i, 16 >> i)
                                                                   for i, d in enumerate(depths))),
            "global_batch": n_gpus * args.batch, "iaf_steps_per_step": n_iaf,
            "iaf_step_samples_per_s": value * n_iaf,
            "weights": "re-derived every step (mask, l2-norm, exp(g)) for all layers in one batched launch" if not args.cached_weights else "prepared once",
            "launch": "hipGraph replay" if graph is not None else "eager",
            "parallelism": "dp%d (batch-sharded replicas, no forward collective)" % n_gpus,
            "live_gflop_per_iaf_step_16x16": work16["live_flops"] / 1e9,
        },
        "roofline": roofline,
    }
    if n_gpus == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, depths)
        out["config"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
