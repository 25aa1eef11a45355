#!/usr/bin/env python3
"""bench.py -- aerial-ground pairs/sec (backbone + ODE fusion + pooling) on N MI355X GPUs.

Workload (BASELINE.json configs[2]/[3], the configuration the metric is quoted on: "synthetic
6x3x224x224 batches"): one PAIR = one 6-camera ground panorama [3,224,1344] through the query
network MM.forward_q (ResNet18 stem+layer1..3, GeM, 3x Neural-ODE fusion blocks, stage-2 fusion)
plus one aerial tile [1,3,224,224] through DBVanilla2D (ResNet18, GeM, MLP) -- SURVEY.md 8(d).
A STEP embeds `--batch` pairs per GPU, then all-gathers the query and database descriptors over
RCCL (the eval path's exchange step); weak scaling: per-GPU work is fixed as N grows.
Steps are replayed hipGraphs; by default TWO steps are in flight (step i on stream i % 2 with its own input batch, workspaces and
outputs: one step's latency-bound tail runs beside the next step's first layers; --inflight 1 for one at a time, and the line
carries that figure too as config.ms_per_step_one_in_flight).  Every step embeds its whole batch; the timed region is K steps
between barrier + synchronize pairs.
Inputs are synthetic (seeded N(0,1) images, U(0,1) voxel stand-ins) and resident in HBM before
the timed region; weights are seeded random init of the reference architecture.

One JSON line on rank 0 (driver contract) with `roofline` (dominant kernel = the implicit-GEMM
conv, timed live with events on the launch stream) and `cpu_baseline` (the CPU oracle = a port of
the reference forward, on the host cores, bounded sample).  Also reports kNN queries/s at
100k x 256 (the second half of BASELINE's metric string) under "knn".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench_inputs  # noqa: E402

from bench_legs import (PEAK_BF16_DENSE_TFLOPS, PROFILE_TAG, conv_roofline, cpu_baseline_measurement,  # noqa: E402,F401
                        default_precision_leg, knn_distributed_leg, knn_measurement, knn_parity, reference_dependency_rows,
                        netvlad_leg, train_measurement, vox_leg)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--windows", type=int, default=5,
                    help="timed windows of --steps steps each (every one bracketed by barrier + synchronize); the line reports the "
                         "MEDIAN window, config.windows carries all of them")
    ap.add_argument("--batch", type=int, default=64, help="pairs per GPU per step (SURVEY.md 8(d): b in {16, 32, 64} per GPU)")
    ap.add_argument("--prec", type=int, default=0, choices=[0, 2, 3, 4],
                    help="MFMA precision of the convs: 0 (default) = the library default, Options().mfma_precision (4 = fp16 x fp16, "
                         "one product; descriptors <= 4e-4 vs fp32); 2 = the opt-in tight mode: fp16 activations x fp16 hi + e4m3 lo "
                         "weights (descriptors 3e-5..1.6e-4, maps <= 6e-4); 3 = split-bf16 (~1e-5)")
    ap.add_argument("--lo-fp8", type=int, default=1, choices=[0, 1],
                    help="--prec 2: the weight-residual (lo) product of the 3x3 convs on the block-scaled e4m3 MFMA "
                         "(agp_conv_desc.w_q8; same accuracy, 3/4 of the MFMA work); 0 = fp16 lo product")
    ap.add_argument("--graph", type=int, default=1, help="replay the step from a captured hipGraph")
    ap.add_argument("--streams", type=int, default=2, choices=[1, 2],
                    help="2 = the database network runs on a second HIP stream next to the query network "
                         "(its small launches fill the tails of the query network's kernels)")
    ap.add_argument("--inflight", type=int, default=0, choices=[0, 1, 2, 3, 4],
                    help="0 = auto (2 on resident inputs, 1 with --h2d: under the upload the overlap buys nothing).  N > 1: N steps in flight -- step i replays its own captured graph on stream i %% N (own input batch, "
                         "workspaces and outputs), so the latency-bound tail of one step (vector programs, the last small launches) "
                         "runs beside the next step's stem and layer 1; every step still embeds its whole batch.  The line also "
                         "carries the same graph's time with one step in flight (config.ms_per_step_one_in_flight).  Applies to the "
                         "graph-replayed step on resident inputs (not --h2d / --vox / --graph 0)")
    ap.add_argument("--qsplit", type=int, default=0, choices=[0, 1, 2, 4],
                    help="N > 1: the query batch is embedded as N equal sub-batches on N HIP streams (same work per step; "
                         "one sub-batch's kernel tails overlap the others' kernels)")
    ap.add_argument("--config", type=str, default="c3", choices=["c3", "c2"],
                    help="c3 (default, the configuration BASELINE.json's metric is quoted on): nuScenes-AG 6-camera panorama, ResNet18 "
                         "trunks, euler h=0.1; c2: KITTI-360-AG cam00 -- one 224x224 ground image through MM with the 4-step RK4 (3/8) "
                         "solver, database tiles through DBVanilla2D with a ResNet50 trunk (prints its own line)")
    ap.add_argument("--pair", type=int, default=1, choices=[0, 1],
                    help="1 = the query and the database trunk advance in lock-step and every layer's 3x3 convs are ONE grouped "
                         "launch (agplace_amd.pair.embed_pair); 0 = the two models are called separately (--streams applies)")
    ap.add_argument("--u8", action="store_true",
                    help="query images enter as uint8 camera tiles [b,6,224,224,3] (device-side normalise + concat + pack) "
                         "instead of the normalised fp32 panorama the reference's model boundary takes")
    ap.add_argument("--h2d-depth", type=int, default=2, help="slots of the pinned ring (--h2d)")
    ap.add_argument("--h2d", action="store_true",
                    help="the step INCLUDES moving its inputs from pinned host memory: decoded uint8 camera tiles and aerial tiles go "
                         "through a 2-slot pinned ring (agplace_amd.input_pipeline.PinnedRing), the upload of step i+1 runs on a copy "
                         "stream under the compute of step i (reference: data_dict[k].to(device) at the top of the step, train.py:303-304)")
    ap.add_argument("--h2d-in-graph", action="store_true",
                    help="with --h2d: the upload of the NEXT step's slot is a memcpy node of the step's own hipGraph (forked beside the "
                         "compute nodes of the current slot) -- one host submission per step instead of graph launch + copy + two events")
    ap.add_argument("--vox", action="store_true",
                    help="the query network runs its sparse-voxel branch from coords / features (reference mm.py:86-93: MinkFPN, MinkGeM, "
                         "the sparse side of stage 2) on --vox-points voxels per sample instead of taking the branch's outputs as dense "
                         "stand-ins; prints its own line, with the stand-in step of the same run beside it")
    ap.add_argument("--vox-points", type=int, default=8000)
    ap.add_argument("--vox-leg", type=int, default=1, choices=[0, 1],
                    help="1 (default): the line also carries `vox` = the same step end to end from coords / features (the sparse-voxel "
                         "branch inside MM.forward_q), four steps in flight, same models")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-steps", type=int, default=10,
                    help="timed steps of the secondary training measurement (0 = skip): forward + backward + fused Adam on "
                         "8 queries x (panorama + 11 aerial tiles of 256^2) per GPU, the reference's step loss")
    ap.add_argument("--train-vox", type=int, default=1, choices=[0, 1],
                    help="also time the training step with the sparse-voxel branch trained from coords -> train.with_voxel_branch")
    ap.add_argument("--train-fast", type=int, default=1, choices=[0, 1],
                    help="also time the training step in the opt-in fast mode (Options.train_precision = 16) -> train.fast_mode")
    ap.add_argument("--sync-bn", action="store_true",
                    help="training leg, N > 1: synchronised BatchNorm (parallel.enable_sync_batchnorm: global-batch statistics, "
                         "one small all-reduce per BatchNorm layer and direction) instead of per-rank statistics")
    ap.add_argument("--no-knn", action="store_true")
    ap.add_argument("--no-netvlad", action="store_true")
    ap.add_argument("--default-prec-leg", type=int, default=1, choices=[0, 1],
                    help="also time the step in the opt-in tight mode (Options(mfma_precision=2), F16W2) -> config.tight_mode_f16w2; the "
                         "headline IS the library default")
    ap.add_argument("--verbose", action="store_true", help="per-conv-launch table on stderr")
    ap.add_argument("--cpu-pairs", type=int, default=256, help="pairs in the bounded CPU sample (about 10 s of host work)")
    ap.add_argument("--cpu-knn-queries", type=int, default=512, help="queries of the bounded CPU kNN sample")
    return ap.parse_args()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python3 bench.py --gpus N`: this process becomes the launcher.  It has not touched the GPU (importing
        # torch does not) and never will: it starts N child ranks, relays rank 0's JSON line, exits with the worst rc.
        from agplace_amd import launcher
        sys.exit(launcher.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    from agplace_amd import _lib, ops, pair, parallel, retrieval
    from agplace_amd.models_baseline.dbvanilla2d import DBVanilla2D
    from agplace_amd.network_mm.mm import MM
    from agplace_amd.options import Options

    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:           # strict: checked BEFORE the rendezvous, which would otherwise wait for absent ranks
        if int(os.environ.get("RANK", "0")) == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: run plain `python3 bench.py --gpus {args.gpus}` (it starts its "
                  f"own ranks) or `python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus}`",
                  file=sys.stderr)
        sys.exit(2)
    rank, world, local = parallel.init_from_env()
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    # AGP_LOCAL_DEVICE: every rank on that one GPU (control-flow smoke test with AGP_DIST_BACKEND=gloo; never a measurement)
    dev = torch.device("cuda", int(os.environ.get("AGP_LOCAL_DEVICE", local)))
    torch.cuda.set_device(dev)
    _lib.load()
    rccl = None
    if world > 1:
        # proof that the collective library saw every rank: an all-reduce of ones on the compute device
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        rccl = {"backend": dist.get_backend(), "world": dist.get_world_size(), "allreduce_ones": float(ones.item()),
                "devices": sorted(set(parallel.all_gather_object(torch.cuda.current_device())))}

    c2 = args.config == "c2"
    if args.prec == 0:
        args.prec = Options().mfma_precision        # the headline measures what `MM()` does without options
    opt = Options(mfma_precision=args.prec, dbimage_fe="resnet50", dbimage_fe_layers="3_4_6", odeint_method="rk4",
                  odeint_size=0.25) if c2 else Options(mfma_precision=args.prec)
    qw = 224 if c2 else 1344            # c2: one camera; c3: six 224-pixel camera tiles concatenated along the width
    ops.LO_FP8 = bool(args.lo_fp8)
    torch.set_grad_enabled(False)       # inference forward (reference test.py:121 runs under no_grad)
    torch.manual_seed(0)
    modelq = MM(opt=opt).to(dev).eval()
    modeldb = DBVanilla2D("db", opt.features_dim, opt=opt).to(dev).eval()
    b = args.batch
    data = bench_inputs.synth_query(b, 224, qw, opt, seed=100 + rank)
    data = {k: ([t.to(dev) for t in v] if isinstance(v, list) else v.to(dev)) for k, v in data.items()}
    if args.u8:
        data["query_image"] = torch.randint(0, 256, (b, qw // 224, 224, 224, 3), dtype=torch.uint8,
                                            generator=torch.Generator().manual_seed(100 + rank)).to(dev)
    tiles = torch.randn(b, 1, 3, 224, 224, generator=torch.Generator().manual_seed(200 + rank)).to(dev)
    data_standins = data
    if args.vox:
        coords, feats = bench_inputs.synth_cloud_lidar(b, args.vox_points, seed=400 + rank)
        data = {k: v for k, v in data.items() if k not in ("vox_levels", "voxfeatvec", "stg2voxvec", "voxvec_fuse")}
        data["coords"], data["features"] = coords.to(dev), feats.to(dev)

    if args.inflight == 0:
        # (with the sparse-voxel branch a step is a long chain of ~55 small kernels, several of them 64 workgroups wide: a third and
        # a fourth step in flight fill what two leave idle -- 3.00-3.04 ms with three, 2.93-3.00 with four on one box)
        args.inflight = 1 if args.h2d else (4 if args.vox else 2)
    ring = None
    # --h2d with steps in flight: slot s replays on stream s % inflight, and a slot is refilled only after the step that read it:
    # twice as many slots as steps in flight keep every stream's next upload ahead of it
    ring_flight = args.inflight if (args.h2d and args.graph and not args.h2d_in_graph and args.inflight > 1) else 1
    if ring_flight > 1:
        args.h2d_depth = max(args.h2d_depth, 2 * ring_flight)
    if args.h2d:
        from agplace_amd.input_pipeline import PinnedRing
        ring = PinnedRing({"q": ((b, qw // 224, 224, 224, 3), torch.uint8), "t": ((b, 1, 224, 224, 3), torch.uint8)}, depth=args.h2d_depth, device=dev)
        gh = torch.Generator().manual_seed(300 + rank)
        for s_ in range(ring.depth):            # the "dataloader": every slot holds a different decoded batch
            for name, t in ring.host(s_).items():
                t.copy_(torch.randint(0, 256, t.shape, dtype=torch.uint8, generator=gh))
    side = torch.cuda.Stream(device=dev) if args.streams == 2 else None
    # sub-batches of a step: 2 when one step is in flight at a time (its own halves overlap each other's tails); with several steps
    # in flight the steps overlap each other and whole-batch launches are the more efficient ones (1.72 against 1.78 ms)
    flight_ok = args.inflight > 1 and args.graph and not args.h2d and args.pair
    if args.qsplit == 0:
        args.qsplit = 1 if (flight_ok or ring_flight > 1) else 2
    nq_s = args.qsplit if (b % args.qsplit == 0 and b >= 2 * args.qsplit) else 1
    opt.query_substreams = nq_s          # MM.forward embeds the batch as nq_s sub-batches on nq_s streams

    def embed_q():
        return modelq(data, mode="q")["embedding"]

    def embed(serial=False, slot=None, dq=None, tl=None):
        dq_ = data if dq is None else dq
        tiles_ = tiles if tl is None else tl
        if slot is not None:                        # --h2d: this slot's device tensors (static addresses: capturable)
            dv = ring.device_tensors(slot)
            dq = dict(data)
            dq["query_image"] = dv["q"]
            oq, od = pair.embed_pair(modelq, modeldb, dq, {"db_map": dv["t"]})
            return oq["embedding"], od["embedding"]
        if args.pair:
            # query and database trunks in lock-step: every layer's 3x3 convs as ONE grouped launch (agplace_amd/pair.py)
            if serial:
                opt.query_substreams = 1
            try:
                oq, od = pair.embed_pair(modelq, modeldb, dq_, {"db_map": tiles_})
            finally:
                opt.query_substreams = nq_s
            return oq["embedding"], od["embedding"]
        if serial:
            opt.query_substreams = 1
            try:
                eq = modelq(data, mode="q")["embedding"]
            finally:
                opt.query_substreams = nq_s
            ed = modeldb({"db_map": tiles}, mode="db")["embedding"]
            return eq, ed
        if side is None:
            return embed_q(), modeldb({"db_map": tiles}, mode="db")["embedding"]
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)                       # fork
        with torch.cuda.stream(side):
            ed = modeldb({"db_map": tiles}, mode="db")["embedding"]
        eq = embed_q()
        cur.wait_stream(side)                       # join
        ed.record_stream(cur)
        return eq, ed

    def exchange(eq, ed):
        if world > 1:
            both = torch.cat([eq, ed], 1)
            return parallel.all_gather_rows(both, equal=True)     # every rank embeds `--batch` pairs
        return None

    # ---- optional hipGraph of the embedding part (static shapes; collectives stay outside)
    graph = None
    eq = ed = None
    # Workspaces are keyed by the launching stream: warm up on the stream the capture will use, so that
    # the capture itself allocates (and zero-fills) nothing.
    cap_stream = torch.cuda.Stream(device=dev)
    cap_stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap_stream):
        for _ in range(2):            # eager warm-up: builds weight planes and workspaces
            eq, ed = embed()
    torch.cuda.synchronize()
    if args.graph:
        try:
            graph = torch.cuda.CUDAGraph()
            # thread_local: the RCCL watchdog thread of a multi-rank run must not invalidate the capture
            with torch.cuda.graph(graph, stream=cap_stream, capture_error_mode="thread_local"):
                eq, ed = embed()
        except Exception as e:          # keep going eagerly, but say so
            graph = None
            if rank == 0:
                print(f"bench.py: hipGraph capture failed ({e!r}); running eagerly", file=sys.stderr)
            torch.cuda.synchronize()
    if graph is None:
        for _ in range(2):            # eager mode runs on the default stream: its own workspaces
            eq, ed = embed()
        torch.cuda.synchronize()

    # ---- --inflight N: one more captured graph per extra in-flight step, each on a stream (and therefore workspaces) of its own,
    # each with its OWN input batch (the steps in flight embed different data, as a dataset pass would)
    flight = None
    if args.inflight > 1 and graph is not None and ring is None and args.pair:
        flight = [(cap_stream, graph, (eq, ed), (data, tiles))]
        for k_ in range(1, args.inflight):
            dk = bench_inputs.synth_query(b, 224, qw, opt, seed=100 + rank + 1000 * k_)
            dk = {k: ([t.to(dev) for t in v] if isinstance(v, list) else v.to(dev)) for k, v in dk.items()}
            if args.vox:
                ck, fk = bench_inputs.synth_cloud_lidar(b, args.vox_points, seed=400 + rank + 1000 * k_)
                dk = {k: v for k, v in dk.items() if k not in ("vox_levels", "voxfeatvec", "stg2voxvec", "voxvec_fuse")}
                dk["coords"], dk["features"] = ck.to(dev), fk.to(dev)
            if args.u8:
                dk["query_image"] = torch.randint(0, 256, (b, qw // 224, 224, 224, 3), dtype=torch.uint8,
                                                  generator=torch.Generator().manual_seed(100 + rank + 1000 * k_)).to(dev)
            tk = torch.randn(b, 1, 3, 224, 224, generator=torch.Generator().manual_seed(200 + rank + 1000 * k_)).to(dev)
            st_ = torch.cuda.Stream(device=dev)
            st_.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st_):
                for _w in range(2):
                    o_ = embed(dq=dk, tl=tk)
            torch.cuda.synchronize()
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_, stream=st_, capture_error_mode="thread_local"):
                o_ = embed(dq=dk, tl=tk)
            flight.append((st_, g_, o_, (dk, tk)))
        torch.cuda.synchronize()

    graphs2 = None
    if ring is not None:
        # one captured graph per ring slot (a graph bakes in the addresses of the device tensors it reads)
        slot_streams = [cap_stream] + [torch.cuda.Stream(device=dev) for _ in range(ring_flight - 1)]
        for st_ in slot_streams[1:]:
            st_.wait_stream(torch.cuda.current_stream())
        for _w in range(2):
            for s_ in range(ring.depth):
                with torch.cuda.stream(slot_streams[s_ % ring_flight]):
                    ring.upload(s_)
                    ring.acquire(s_)
                    embed(slot=s_)
        torch.cuda.synchronize()
        if args.graph:
            graphs2 = []
            for s_ in range(ring.depth):
                gph = torch.cuda.CUDAGraph()
                cap_s = slot_streams[s_ % ring_flight]
                with torch.cuda.graph(gph, stream=cap_s, capture_error_mode="thread_local"):
                    if args.h2d_in_graph:
                        # the copy of the NEXT slot as a node of this graph, on a forked stream beside this slot's compute
                        nxt = (s_ + 1) % ring.depth
                        ring.copy_stream.wait_stream(cap_s)
                        with torch.cuda.stream(ring.copy_stream):
                            ring.device_arena(nxt).copy_(ring.host_arena(nxt), non_blocking=True)
                    outs_ = embed(slot=s_)
                    if args.h2d_in_graph:
                        cap_s.wait_stream(ring.copy_stream)
                graphs2.append((gph, outs_))
        for s_ in range(ring.depth):
            ring.upload(s_)
    step_no = [0]
    ex_done = [None] * 8      # steps in flight on several ranks: the event after the all-gather that read a slot's outputs

    def step():
        nonlocal eq, ed
        if ring is not None and args.h2d_in_graph and graphs2 is not None:
            s_ = step_no[0] % ring.depth
            step_no[0] += 1
            graphs2[s_][0].replay()                 # computes slot s_ and uploads slot s_ + 1 (filled by the host meanwhile)
            eq, ed = graphs2[s_][1]
        elif ring is not None:
            s_ = step_no[0] % ring.depth
            step_no[0] += 1
            st_ = slot_streams[s_ % ring_flight] if ring_flight > 1 else torch.cuda.current_stream()
            if world > 1 and ring_flight > 1 and ex_done[s_ % len(ex_done)] is not None:
                st_.wait_event(ex_done[s_ % len(ex_done)])
            with torch.cuda.stream(st_):
                ring.acquire(s_)                    # the slot's stream waits for this slot's upload (issued `depth` steps ago)
                if graphs2 is not None:
                    graphs2[s_][0].replay()
                    eq, ed = graphs2[s_][1]
                else:
                    eq, ed = embed(slot=s_)
                ring.release(s_)
                ring.upload(s_)                     # refill for step i + depth: runs on the copy stream under the next steps
            if world > 1 and ring_flight > 1:
                cur_ = torch.cuda.current_stream()
                cur_.wait_stream(st_)
                exchange(eq, ed)
                ex_done[s_ % len(ex_done)] = torch.cuda.Event()
                ex_done[s_ % len(ex_done)].record(cur_)
                return
        elif flight is not None:
            k_ = step_no[0] % len(flight)
            st_, g_, (eq, ed), _in = flight[k_]
            step_no[0] += 1
            if world > 1 and ex_done[k_] is not None:
                st_.wait_event(ex_done[k_])          # this slot's previous outputs have been read by their all-gather
            with torch.cuda.stream(st_):
                g_.replay()
            if world > 1:
                cur_ = torch.cuda.current_stream()
                cur_.wait_stream(st_)
                exchange(eq, ed)
                ex_done[k_] = torch.cuda.Event()
                ex_done[k_].record(cur_)
            return
        elif graph is not None:
            graph.replay()
        else:
            eq, ed = embed()
        exchange(eq, ed)

    for _ in range(args.warmup):
        step()

    def timed_window():
        """EXACTLY --steps steps between barrier + synchronize pairs; MAX over ranks."""
        parallel.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        parallel.barrier()
        w = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([w], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            w = float(tt.item())
        return w
    # a window of --steps steps is tens of milliseconds: one host hiccup or a clock step of the box moves it by more than a round's
    # kernel gain.  `value` is the MEDIAN of --windows such windows (each one the contract's timed region), min / max in config.
    window_s = [timed_window() for _ in range(max(1, args.windows))]
    dt = sorted(window_s)[len(window_s) // 2]
    pairs_per_s = world * b * args.steps / dt
    ms_one_in_flight = None
    if flight is not None:       # the same graph with ONE step in flight (each replay waits for the previous one), same run
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        with torch.cuda.stream(cap_stream):
            for _ in range(args.steps):
                graph.replay()
        torch.cuda.synchronize()
        ms_one_in_flight = round((time.perf_counter() - t1) / args.steps * 1e3, 3)

    # ---- the same step in the opt-in tight mode (F16W2), same run (the headline is the library default since round 5)
    library_default = None
    if args.prec == 4 and flight is not None and not args.vox and not c2 and not args.u8 and args.default_prec_leg:
        try:
            library_default = default_precision_leg(args, dev, [f_[3] for f_ in flight], world, b, pair, MM, DBVanilla2D, Options)
        except Exception as e:
            if rank == 0:
                print(f"bench.py: default-precision leg failed: {e!r}", file=sys.stderr)

    # ---- --vox: the same step with the voxel branch's outputs as dense stand-ins (what the headline line runs), same run, same
    # models, captured and timed the same way: what the branch adds
    vox_cmp = None
    if args.vox and args.pair and ring is None:
        with torch.cuda.stream(cap_stream):
            for _ in range(2):
                embed(dq=data_standins)
        torch.cuda.synchronize()
        g2 = None
        if graph is not None:
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, stream=cap_stream, capture_error_mode="thread_local"):
                embed(dq=data_standins)
        for _ in range(args.warmup):
            g2.replay() if g2 is not None else embed(dq=data_standins)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            g2.replay() if g2 is not None else embed(dq=data_standins)
        torch.cuda.synchronize()
        ms_st = (time.perf_counter() - t1) / args.steps * 1e3
        # (like with like: both with ONE step in flight when the timed region ran several)
        ms_vx = ms_one_in_flight if ms_one_in_flight is not None else dt / args.steps * 1e3
        vox_cmp = {"ms_per_step_with_voxel_branch": round(ms_vx, 3), "ms_per_step_dense_standins": round(ms_st, 3),
                   "voxel_branch_adds": round(ms_vx / ms_st - 1.0, 4), "voxels_requested_per_sample": args.vox_points,
                   "voxel_coords_in_range": modelq.voxel_coords_in_range()}

    # ---- outside the timed region: the replayed hipGraph computes what the eager path computes (bit for bit: the same
    # kernels on the same buffers; None when the step was not a graph replay of resident inputs)
    replay_equals_eager = None
    if graph is not None and ring is None:
        replay_equals_eager = True
        for st_, g_, (oq_, od_), (dk, tk) in (flight if flight is not None else [(cap_stream, graph, (eq, ed), (data, tiles))]):
            with torch.cuda.stream(st_):
                g_.replay()
            torch.cuda.synchronize()
            rq, rd = oq_.clone(), od_.clone()
            with torch.cuda.stream(st_):
                xq, xd = embed(dq=dk, tl=tk)
            torch.cuda.synchronize()
            replay_equals_eager = replay_equals_eager and bool(torch.equal(rq, xq) and torch.equal(rd, xd) and torch.isfinite(rq).all()
                                                               and float(rq.abs().sum()) > 0)

    roofline = conv_roofline(args, embed, ops, rank, c2)

    # ---- the step END TO END from coords (the sparse-voxel branch inside MM.forward_q) in the default line, same models
    vox = None
    if args.vox_leg and not args.vox and not c2 and not args.u8 and ring is None and graph is not None and args.pair:
        try:
            vox = vox_leg(args, embed, modelq, data_standins, b, qw, opt, dev, rank, world)
        except Exception as e:
            if rank == 0:
                print(f"bench.py: voxel-branch leg failed: {e!r}", file=sys.stderr)

    out = {
        "metric": "aerial-ground pairs/sec (backbone+ODE+pool)", "value": round(pairs_per_s, 2),
        "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": {2: ("f16w2 (fp16 activations x fp16 hi + e4m3 lo weights; 3x3 convs: f16 MFMA + block-scaled fp8 MFMA for the "
                                           "lo product, fp32 accumulate)" if args.lo_fp8 else
                                           "f16w2 (fp16 activations x fp16 hi+lo weights, 2 MFMA products, fp32 accumulate)"),
                                       3: "bf16x3 (split-bf16 MFMA, fp32 accumulate)",
                                       4: "f16 (fp16 x fp16, fp32 accumulate)"}[args.prec],
        "data": "synthetic",
        "config": {"workload": ("KITTI-360-AG cam00 (C2): MM.forward_q on [b,3,224,224] (ResNet18 stem+layer1-3, 4-step RK4 3/8-rule x3 "
                                "FCODE, GeM, stage-2 fusion) + DBVanilla2D on [b,1,3,224,224] with a ResNet50 stem+layer1-3 trunk "
                                "(network/image_fe.py:47-59); inference forward") if c2 else
                               ("nuScenes-AG 6-cam (C3/C4): MM.forward_q on [b,3,224,1344] + DBVanilla2D on "
                                "[b,1,3,224,224], ResNet18 stem+layer1-3, euler h=0.1 x3 FCODE, GeM, stage-2 fusion; "
                                "inference forward" + ("; the sparse-voxel branch (MinkFPN 64-128-256 + ECA blocks + MinkGeM + stage-2 sparse "
                                                       "side) runs from coords / features, reference mm.py:86-93" if args.vox else
                                                       "; the voxel branch's pooled outputs enter as fixed tensors (SURVEY.md 8d)")),
                   "windows": {"n": len(window_s), "steps_per_window": args.steps, "reported": "median",
                               "ms_per_step_min": round(min(window_s) / args.steps * 1e3, 3),
                               "ms_per_step_max": round(max(window_s) / args.steps * 1e3, 3),
                               "ms_per_step_all": [round(w_ / args.steps * 1e3, 3) for w_ in window_s]},
                   "pairs_per_gpu_per_step": b, "global_batch": b * world, "parallelism": f"dp{world}", "bn": "per-rank (eval: running statistics)",
                   "paired_trunks": bool(args.pair),
                   "hipgraph": graph is not None, "replay_equals_eager": replay_equals_eager, "steps_in_flight": len(flight) if flight else (ring_flight if ring is not None else 1),
                   "ms_per_step_one_in_flight": ms_one_in_flight,
                   "library_default_precision": Options().mfma_precision, "headline_is_library_default": args.prec == Options().mfma_precision,
                   "tight_mode_f16w2": library_default,
                   "streams": args.streams, "query_sub_batches_on_streams": nq_s,
                   "query_input": ("uint8 camera tiles + uint8 aerial tiles from PINNED HOST memory inside the step (2-slot ring, "
                                   + ("the next slot's upload is a memcpy node of the step's hipGraph; " if args.h2d_in_graph else "copy stream; ")
                                   + f"{ring.bytes_per_batch / 1e6:.1f} MB per step)") if ring is not None else
                                  ("uint8 camera tiles" if args.u8 else "fp32 normalised panorama"),
                   "gmac_per_pair": round((bench_inputs.resnet_gmacs("resnet18", 3, 224, qw) + bench_inputs.resnet_gmacs(opt.dbimage_fe, 3, 224, 224)
                                           + 14 * (qw // 16) * 256 * 256 * 9 * 2) / 1e9, 3)},
        "roofline": roofline,
    }
    if rccl is not None:
        out["rccl"] = rccl
    if vox_cmp is not None:
        out["voxel_branch"] = vox_cmp
    if vox is not None:
        out["vox"] = vox

    # ---- kNN queries/s at DB = 100k x 256 (query shard per rank, database replicated)
    if not args.no_knn and not c2 and not args.vox:
        out["knn"] = knn_measurement(args, opt, dev, rank, world, parallel, retrieval)

    # ---- secondary metric: training step = fwd + bwd + Adam, 1 query + 11 tiles per "query"
    if args.train_steps > 0 and not c2 and not args.vox:
        try:
            out["train"] = train_measurement(args, opt, dev, rank, world, parallel, side=side)
        except Exception as e:      # never lose the headline line over the secondary metric
            if rank == 0:
                print(f"bench.py: training measurement failed: {e!r}", file=sys.stderr)
        if args.train_fast and "train" in out:
            # the opt-in fast mode (Options.train_precision = 16: one-product forward convs), same step, same run
            try:
                tfm = train_measurement(args, opt.copy(train_precision=16), dev, rank, world, parallel, side=side)
                out["train"]["fast_mode"] = {
                    "value": tfm["value"], "unit": tfm["unit"], "ms_per_step": tfm["ms_per_step"],
                    "speedup_over_tight_mode": round(out["train"]["ms_per_step"] / tfm["ms_per_step"], 3),
                    "dtype": "forward 3x3 convs (stride 1 and 2) and 1x1 stride-2 downsamples: fp16 x fp16, one MFMA product; data gradients bf16x3; weight gradients one fp16 product; "
                             "fp32 master weights, fp64-finalised BatchNorm statistics",
                    "gradient_accuracy": "ResNet18 trunk, every parameter gradient vs fp64 autograd: rel-L2 median 3.1e-3, worst 4.4e-3, "
                                         "cosine >= 0.99999 (tests/test_gpu_train.py::test_resnet_trunk_training_gradients[fast]); the "
                                         "tight mode holds 1e-3",
                    "opt_in": "Options(train_precision=16); the default (32) is the tight mode measured above"}
                # ... and with the data gradients of the 3x3 convs as one bf16 product as well (Options.train_dgrad_products = 1)
                tfd = train_measurement(args, opt.copy(train_precision=16, train_dgrad_products=1), dev, rank, world, parallel, side=side)
                out["train"]["fast_mode_dgrad1"] = {
                    "value": tfd["value"], "unit": tfd["unit"], "ms_per_step": tfd["ms_per_step"],
                    "speedup_over_tight_mode": round(out["train"]["ms_per_step"] / tfd["ms_per_step"], 3),
                    "dtype": "as fast_mode, and the 3x3 data gradients as ONE bf16 x bf16 product of the hi planes (agp_conv_desc.hi_only); "
                             "1x1 / stem gradients and every stored map unchanged (split-bf16 pairs)",
                    "gradient_accuracy": "ResNet18 trunk, every parameter gradient vs fp64 autograd: rel-L2 median 5.1e-3, worst 7.9e-3, "
                                         "cosine >= 0.99996 (tests/test_gpu_train.py::test_resnet_trunk_training_gradients[dgrad1])",
                    "opt_in": "Options(train_precision=16, train_dgrad_products=1)"}
            except Exception as e:
                if rank == 0:
                    print(f"bench.py: fast-mode training measurement failed: {e!r}", file=sys.stderr)
        if args.train_vox and "train" in out:
            try:
                tv = train_measurement(args, opt, dev, rank, world, parallel, side=side, with_coords=True)
                out["train"]["with_voxel_branch"] = {k: tv[k] for k in ("value", "unit", "ms_per_step", "metric", "grad_exchange")}
            except Exception as e:
                if rank == 0:
                    print(f"bench.py: training measurement with the voxel branch failed: {e!r}", file=sys.stderr)

    # ---- NetVLAD.forward on the matrix pipe (the other aggregator of SURVEY.md 8 row a9), this rank's clock
    if rank == 0 and not c2 and not args.vox and not args.no_netvlad:
        try:
            out["netvlad"] = netvlad_leg(dev)
        except Exception as e:
            print(f"bench.py: NetVLAD leg failed: {e!r}", file=sys.stderr)

    # ---- CPU baseline: the oracle (a port of the reference forward) on the host cores, rank 0, N=1
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_measurement(args, opt, modelq, modeldb, data, tiles, b)

    # ---- "reference-dependency CPU" rows: torchdiffeq.odeint and faiss.IndexFlatL2 on
    # the same tensors IF the box's own site-packages have them (reference call sites network_mm/ffns.py:84-85, test.py:27-32)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["reference_deps"] = reference_dependency_rows(opt, args)

    if rank == 0:
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()
    if out.get("knn", {}).get("parity_failed"):
        sys.exit(3)         # a search that returns wrong neighbours: the line says so (knn.value null) and the run fails


if __name__ == "__main__":
    main()
