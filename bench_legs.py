"""The measurement legs of bench.py (the driver contract -- argument parsing, the timed step loop, the one JSON line -- stays in
bench.py; these functions are imported there): the training step, the roofline of the dominant kernel family, the kNN legs
(single GPU and the sharded-database N > 1 leg), the opt-in tight-precision step, the CPU baseline and the reference-dependency rows.
`oracle` is imported only inside cpu_baseline_measurement / knn_measurement's CPU port / knn_parity: checker legs outside every timed
region."""
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

PEAK_BF16_DENSE_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PROFILE_TAG = "r06"                 # profiles/<tag>_pmc_*.json: the committed rocprofv3 PMC summaries the line quotes HBM traffic from


def train_measurement(args, opt, dev, rank, world, parallel, side=None, with_coords=False):
    """forward + backward + fused Adam of MM + DBVanilla2D with the reference's step loss
    (train.py:303-341), gradients all-reduced over RCCL when N > 1; see tools/train_bench.py."""
    import types
    from agplace_amd import losses
    from agplace_amd.models_baseline.dbvanilla2d import DBVanilla2D
    from agplace_amd.network_mm.mm import MM
    torch.set_grad_enabled(True)
    try:
        bq, ndb, tile = 16, 11, 256           # the reference's default train_batch_size (tools/options.py:35) and 1 + 10 tiles per query
        torch.manual_seed(1)
        mq = MM(opt=opt).to(dev).train()
        mdb = DBVanilla2D("db", opt.features_dim, opt=opt).to(dev).train()
        data = bench_inputs.synth_query(bq, 224, 1344, opt, seed=500 + rank)
        data = {k: ([t.to(dev) for t in v] if isinstance(v, list) else v.to(dev)) for k, v in data.items()}
        if with_coords:
            # the voxel branch TRAINED from coords / features (reference train.py:308 -> mm.py:86-93) instead of fed as fixed tensors
            coords, feats = bench_inputs.synth_cloud_lidar(bq, args.vox_points, seed=700 + rank)
            data = {k: v for k, v in data.items() if k not in ("vox_levels", "voxfeatvec", "stg2voxvec", "voxvec_fuse")}
            data["coords"], data["features"] = coords.to(dev), feats.to(dev)
        gen = torch.Generator().manual_seed(600 + rank)
        data["query_eastnorth"] = (torch.rand(bq, 2, generator=gen) * 60).to(dev)
        data["db_eastnorth"] = (torch.rand(bq, ndb, 2, generator=gen) * 60).to(dev)
        nmap = len(opt.maptype.split("_"))
        db = {"db_map": torch.randn(bq, ndb, nmap, 3, tile, tile, generator=gen).to(dev)}
        per, negs = 1 + ndb, ndb - 1
        trip = torch.tensor([[per * i, per * i + 1, per * i + 2 + j] for i in range(bq) for j in range(negs)]).to(dev)
        largs = types.SimpleNamespace(criterion="triplet", train_batch_size=bq, negs_num_per_query=negs, margin=opt.margin)
        # (GradBuckets learns the bucket order from the first step's ready order and drops the parameters no rank has a
        # gradient for -- here the voxel side, whose outputs enter as fixed tensors -- from the exchange)
        named = [("db." + n, p) for n, p in mdb.named_parameters()] + [("q." + n, p) for n, p in mq.named_parameters()]
        named = [(n, p) for n, p in named if p.requires_grad]
        params = [p for _, p in named]
        # the reference's own optimiser layout (train.py:165-190, 213-214): Adam over the database model's one group + Adam over
        # the query model's sixteen groups at lr / lrpc
        from agplace_amd.train_fns import reference_optimizers
        optim_db, optim_q = reference_optimizers(mdb, mq, fused=True)
        # N > 1: one flat gradient buffer the .grad tensors view, all-reduced in buckets while backward still runs
        # (force_buckets: tests/helpers/rccl_single_rank.py runs this step in a ONE-rank RCCL group with the exchange switched on)
        buckets = parallel.GradBuckets(params, bucket_mb=16.0, collective_on_single_rank=True, names=[n for n, _ in named]) if (world > 1 or getattr(args, "force_buckets", False)) else None
        sync_bn = bool(args.sync_bn and world > 1)
        if sync_bn:
            parallel.enable_sync_batchnorm()

        # (reuse the inference section's side stream: ROCm multiplexes streams onto a few hardware queues, and
        # a fifth stream object would share the default stream's queue -- no concurrency at all)
        side = side if side is not None else torch.cuda.Stream(device=dev)

        def step(serial=False):
            if buckets is not None:
                buckets.zero_grad()
            else:
                optim_db.zero_grad(set_to_none=True)
                optim_q.zero_grad(set_to_none=True)
            # the database network's forward -- and with it its backward, which autograd runs on the
            # forward's stream -- goes on a second stream next to the query network's
            # (serial: everything on one stream -- the profiled step below, whose per-launch events must bracket one launch each)
            cur = torch.cuda.current_stream()
            if serial:
                fd = mdb(db, mode="db")
            else:
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    fd = mdb(db, mode="db")
            fq = mq(data, mode="q")
            if not serial:
                cur.wait_stream(side)
            q, d = fq["embedding"], fd["embedding"]
            loss = losses.compute_other_loss(fq, fd, data, opt.train_positives_dist_threshold,
                                             opt.val_positive_dist_threshold, opt=opt)
            feats = torch.cat((q.unsqueeze(1), d), dim=1).view(-1, q.shape[-1])
            loss = loss + losses.compute_loss(largs, None, trip, feats) * opt.tripletloss_weight
            loss.backward()
            if buckets is not None:
                buckets.finish()
            optim_db.step()
            optim_q.step()

        for _ in range(3):
            step()
        parallel.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.train_steps):
            step()
        torch.cuda.synchronize()
        parallel.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        ms = dt / args.train_steps * 1e3
        roofline = None
        if dev.type == "cuda" and not with_coords:
            roofline = train_roofline(step, ms, bq, ndb, tile, opt)
        return {"roofline": roofline, "metric": "training queries/sec (forward + backward + Adam; 1 query = 6-cam panorama + 11 aerial tiles 256x256, "
                          "reference step loss" + ("; the sparse-voxel branch trained from coords: %d requested voxels per query, levels "
                                                   "built by the device-side coordinate manager, one read-back of the row counts" % args.vox_points if with_coords else
                                                   "; the voxel branch's outputs enter as fixed tensors") + ")",
                "value": round(world * bq / ms * 1e3, 1), "unit": "queries/s",
                "ms_per_step": round(ms, 3), "queries_per_gpu_per_step": bq, "images_per_s": round(world * bq * per / ms * 1e3, 1),
                "dtype": "bf16x3 (split-bf16 maps and MFMA, fp32 accumulate)", "steps": args.train_steps,
                "bn": ("synchronised: global-batch statistics (parallel.enable_sync_batchnorm)" if sync_bn else
                       "per-rank batch statistics (parallel.sync_bn_buffers before checkpoints)"),
                # what the LAST step's exchange did (parallel.GradBuckets.stats): buckets, how many all-reduces were launched while
                # backward still ran, how many were held back by parameters without a gradient, bytes exchanged / of them zeros
                "grad_exchange": "none (1 rank)" if buckets is None else dict(buckets.stats, bucket_mb=16.0)}
    finally:
        if "buckets" in locals() and buckets is not None:
            buckets.close()
        parallel.enable_sync_batchnorm(None)
        torch.set_grad_enabled(False)


def train_roofline(step, ms_per_step, bq, ndb, tile, opt):
    """`train.roofline`: the training step's dominant kernel family -- the THREE-PRODUCT (split-bf16) forward and data-gradient
    3x3 stride-1 convolutions (agp_igemm::igemm_kxr_kernel; the weight gradients run one fp16 product on kernels of their own) --
    from HIP events on the launch stream around every conv launch of ONE single-stream step (ops.CONV_PROFILE), plus the
    whole step priced against the same peak: algorithmic flop = 3 x the forward's (forward + data gradient + weight gradient)."""
    from agplace_amd import ops
    for _ in range(2):
        step(serial=True)              # the default stream's workspaces
    torch.cuda.synchronize()
    ops.CONV_PROFILE = []
    try:
        step(serial=True)
        torch.cuda.synchronize()
        prof = ops.CONV_PROFILE
    finally:
        ops.CONV_PROFILE = None
    fam = [p for p in prof if p[3][5] == 3 and p[3][6] == 3 and p[3][7] == 1]
    fam_ms = sum(p[0].elapsed_time(p[1]) for p in fam)
    fam_macs = sum(p[2] for p in fam)
    all_ms = sum(p[0].elapsed_time(p[1]) for p in prof)
    all_macs = sum(p[2] for p in prof)
    ach = 2.0 * fam_macs / (max(fam_ms, 1e-9) * 1e-3) / 1e12
    # the whole step: per query one panorama through the query trunk + stage 2 and `ndb` tiles through the database trunk
    fwd_gmac = (bench_inputs.resnet_gmacs("resnet18", 3, 224, 1344) + 14 * 84 * 256 * 256 * 9 * 2
                + ndb * bench_inputs.resnet_gmacs(opt.dbimage_fe, 3, tile, tile)) / 1e9
    step_gflop = bq * fwd_gmac * 3 * 2
    step_tf = step_gflop / ms_per_step
    traffic, note = None, f"null: no profiles/{PROFILE_TAG}_pmc_train.json measured on these kernel sources"
    try:
        with open(os.path.join(ROOT, f"profiles/{PROFILE_TAG}_pmc_train.json")) as f:
            pt = json.load(f)
        if pt.get("csrc_sha16") == bench_inputs.kernel_source_sha16(ROOT):
            ks = [v for k, v in pt.get("kernels", {}).items() if "igemm_kxr_kernel" in k and v.get("hbm_mb_per_launch")]
            if ks:
                n = sum(v["launches_per_step"] for v in ks)
                traffic = round(sum(v["hbm_mb_per_launch"] * v["launches_per_step"] for v in ks) / n * 1e6)
                note = (f"HBM bytes per igemm_kxr launch of a training step (profiles/{PROFILE_TAG}_pmc_train.json: separate rocprofv3 "
                        "--pmc FETCH_SIZE / WRITE_SIZE passes of tools/train_bench.py)")
    except Exception:
        pass
    return {"bound": "mfma",
            "kernel": "agp_igemm::igemm_kxr_kernel -- the forward and data-gradient 3x3 stride-1 convolutions of a step on split-bf16 "
                      "operands (three MFMA products per algorithmic flop: hi*hi + hi*lo + lo*hi)",
            "achieved": round(ach, 2), "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_DENSE_TFLOPS, 4),
            "mfma_passes_per_algorithmic_flop": 3, "issued_frac": round(3 * ach / PEAK_BF16_DENSE_TFLOPS, 4),
            "launches_per_step": len(fam), "avg_launch_ms": round(fam_ms / max(len(fam), 1), 4),
            "algorithmic_gflop_per_launch": round(2.0 * fam_macs / max(len(fam), 1) / 1e9, 3), "kernel_ms_per_step": round(fam_ms, 3),
            "traffic": traffic, "traffic_unit": note,
            "all_forward_and_dgrad_convs": {"launches_per_step": len(prof), "ms_per_step": round(all_ms, 3),
                                            "achieved": round(2.0 * all_macs / (max(all_ms, 1e-9) * 1e-3) / 1e12, 2),
                                            "note": "MACs as executed (a stride-2 data gradient runs on the zero-upsampled gradient)"},
            "whole_step": {"algorithmic_gflop": round(step_gflop, 1), "achieved": round(step_tf, 2),
                           "frac": round(step_tf / PEAK_BF16_DENSE_TFLOPS, 4),
                           "note": "3 x the forward's flop (forward + data gradient + weight gradient) / ms_per_step; the step also holds "
                                   "the BatchNorm / pooling passes (HBM-bound), the vector path and the optimiser"}}


def reference_dependency_rows(opt, args):
    """{"torchdiffeq": false | {...}, "faiss": false | {...}}: the reference's two third-party hot-path dependencies timed
    on the host cores when they can be imported here (they are not part of this image; nothing is installed or stubbed)."""
    rows = {}
    try:
        import torchdiffeq                                     # noqa: F401
        f = torch.nn.Linear(256, 256)
        x = torch.randn(64, 256)

        def field(t, y):
            return torch.relu(f(y))
        with torch.no_grad():
            torchdiffeq.odeint(field, x, torch.tensor([0.0, 1.0]), method=opt.odeint_method, options={"step_size": opt.odeint_size})
            t0 = time.perf_counter()
            for _ in range(20):
                torchdiffeq.odeint(field, x, torch.tensor([0.0, 1.0]), method=opt.odeint_method, options={"step_size": opt.odeint_size})
            dt = (time.perf_counter() - t0) / 20
        rows["torchdiffeq"] = {"version": getattr(torchdiffeq, "__version__", "?"), "fcode_64x256_ms": round(dt * 1e3, 3),
                               "method": opt.odeint_method, "step_size": opt.odeint_size}
    except ImportError:
        rows["torchdiffeq"] = False
    try:
        import faiss
        import numpy as np
        g = torch.Generator().manual_seed(1)
        db = torch.randn(100000, 256, generator=g)
        db = (db / db.norm(dim=1, keepdim=True)).numpy()
        q = torch.randn(4096, 256, generator=g)
        q = (q / q.norm(dim=1, keepdim=True))[:args.cpu_knn_queries].numpy()
        index = faiss.IndexFlatL2(256)
        index.add(np.ascontiguousarray(db))
        index.search(q[:32], 20)
        t0 = time.perf_counter()
        index.search(q, 20)
        dt = time.perf_counter() - t0
        rows["faiss"] = {"version": getattr(faiss, "__version__", "?"), "queries_per_s": round(q.shape[0] / dt, 1),
                         "sample": f"{q.shape[0]} queries, 100k x 256, k=20, IndexFlatL2 on the host cores"}
    except ImportError:
        rows["faiss"] = False
    return rows


def default_precision_leg(args, dev, inputs, world, b, pair, MM, DBVanilla2D, Options):
    """`config.tight_mode_f16w2`: the SAME step in the opt-in tight mode (Options(mfma_precision=2), F16W2: fp16 activations x
    fp16 hi + e4m3 lo weights, 1.5 MFMA passes per algorithmic flop; the library default of rounds 1-4) -- same weights (same
    seed), same input batches, one captured graph per in-flight step, replayed the same way, in the same run.  The headline is
    what `MM()` does without options (Options().mfma_precision = 4 since round 5)."""
    opt2 = Options(mfma_precision=2)
    torch.manual_seed(0)
    mq = MM(opt=opt2).to(dev).eval()
    mdb = DBVanilla2D("db", opt2.features_dim, opt=opt2).to(dev).eval()
    opt2.query_substreams = 1
    fl = []
    for dk, tk in inputs:
        st = torch.cuda.Stream(device=dev)
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            for _ in range(2):
                pair.embed_pair(mq, mdb, dk, {"db_map": tk})
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
            pair.embed_pair(mq, mdb, dk, {"db_map": tk})
        fl.append((st, g))

    def run(n):
        for i in range(n):
            st, g = fl[i % len(fl)]
            with torch.cuda.stream(st):
                g.replay()
    run(args.warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"prec": 2, "dtype": "f16w2 (fp16 activations x fp16 hi + e4m3 lo weights; f16 MFMA + block-scaled fp8 MFMA, fp32 accumulate)",
            "ms_per_step": round(dt / args.steps * 1e3, 3), "pairs_per_s": round(world * b * args.steps / dt, 2),
            "steps_in_flight": len(fl), "note": "opt-in: Options(mfma_precision=2); this rank's clock, no exchange inside"}


def vox_leg(args, embed, modelq, data_standins, b, qw, opt, dev, rank, world, nflight=4, windows=3):
    """`vox` of the default line: the SAME step END TO END from coords / features -- MM.forward_q with its
    sparse-voxel branch (reference mm.py:86-93: MinkFPN, MinkGeM, the sparse side of stage 2) instead of the branch's pooled
    outputs as fixed tensors -- same models, `nflight` captured graphs (own input batch, own stream), replayed round-robin like the
    headline's.  Timed as the median of `windows` windows of --steps steps; the replayed outputs are compared with an eager pass."""
    flight = []
    for k_ in range(nflight):
        dk = bench_inputs.synth_query(b, 224, qw, opt, seed=100 + rank + 1000 * k_)
        dk = {k: ([t.to(dev) for t in v] if isinstance(v, list) else v.to(dev)) for k, v in dk.items()
              if k not in ("vox_levels", "voxfeatvec", "stg2voxvec", "voxvec_fuse")}
        ck, fk = bench_inputs.synth_cloud_lidar(b, args.vox_points, seed=400 + rank + 1000 * k_)
        dk["coords"], dk["features"] = ck.to(dev), fk.to(dev)
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

    def run(n):
        for i in range(n):
            st_, g_, _o, _in = flight[i % len(flight)]
            modelq.poll_voxel_range()        # pinned host words the replays publish: no stream is touched (raises on a bad cloud)
            with torch.cuda.stream(st_):
                g_.replay()
    run(max(args.warmup, nflight))
    ws = []
    for _ in range(windows):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize()
        ws.append(time.perf_counter() - t0)
    dt = sorted(ws)[len(ws) // 2]
    same = True
    for st_, g_, (oq_, od_), (dk, tk) in flight[:2]:
        with torch.cuda.stream(st_):
            g_.replay()
        torch.cuda.synchronize()
        rq, rd = oq_.clone(), od_.clone()
        with torch.cuda.stream(st_):
            xq, xd = embed(dq=dk, tl=tk)
        torch.cuda.synchronize()
        same = same and bool(torch.equal(rq, xq) and torch.equal(rd, xd) and torch.isfinite(rq).all() and float(rq.abs().sum()) > 0)
    in_range = bool(modelq.voxel_coords_in_range())
    del flight
    return {"metric": "aerial-ground pairs/sec END TO END from coords (MM.forward_q with its sparse-voxel branch: MinkFPN 64-128-256 + ECA "
                      "blocks + MinkGeM + the sparse side of stage 2, reference mm.py:86-93) + DBVanilla2D",
            "ms_per_step": round(dt / args.steps * 1e3, 3), "pairs_per_s": round(world * b * args.steps / dt, 2),
            "voxels": {"requested_per_sample": args.vox_points, "samples_per_step": b},
            "steps_in_flight": nflight, "windows": len(ws), "replay_equals_eager": same, "voxel_coords_in_range": in_range,
            "this_rank_clock": world > 1}


PEAK_F32_MATRIX_TFLOPS = 157.3      # v_mfma_f32_32x32x2_f32: fp32 operands, 64 FLOP / clk / SIMD (MI355X_MICROARCH.md)


def netvlad_leg(dev, n=64, d=256, h=14, w=84, k=64, reps=30):
    """`netvlad`: NetVLAD.forward (reference model/aggregation.py:126-146; K = 64 clusters) on a batch of layer-3 maps
    [64, 256, 14, 84] -- the soft-assignment logits [K] x [D] x [hw] and V = a x^T as exact-fp32 MFMA GEMMs, several workgroups per
    image (csrc/netvlad.hip, round 6).  HIP events on the launch stream around `reps` forwards; the one-workgroup-per-image VALU
    kernel of rounds 1-5 timed beside it."""
    from agplace_amd import ops
    from agplace_amd.model.aggregation import NetVLAD
    g = torch.Generator().manual_seed(5)
    m = NetVLAD(clusters_num=k, dim=d).to(dev)
    with torch.no_grad():
        m.conv.weight.copy_((torch.randn(k, d, 1, 1, generator=g) * 2.0).to(dev))
        m.centroids.copy_(torch.randn(k, d, generator=g).to(dev))
    x = torch.randn(n, d, h, w, generator=g).to(dev)

    def timed():
        # replayed from a hipGraph like the inference step (an eager call is ~60 us of host work for two launches of ~40 us)
        st = torch.cuda.Stream(device=dev)
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            for _ in range(3):
                m(x)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st, capture_error_mode="thread_local"):
            y = m(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(st):
            for _ in range(3):
                gr.replay()
            e0.record()
            for _ in range(reps):
                gr.replay()
            e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps, y.clone()
    ms, y = timed()
    ops.NETVLAD_MFMA = False
    try:
        ms_valu, yv = timed()
    finally:
        ops.NETVLAD_MFMA = True
    flop = 2.0 * 2.0 * k * d * h * w * n                     # two GEMMs of k x d x hw per image
    nbytes = (x.numel() + y.numel()) * 4
    tf = flop / (ms * 1e-3) / 1e12
    return {"metric": "NetVLAD.forward images/sec (K = 64, maps [64, 256, 14, 84] fp32 NCHW -> [64, 16384])",
            "value": round(n / ms * 1e3, 1), "unit": "images/s", "ms_per_batch": round(ms, 4), "dtype": "f32 (fp32 x fp32 MFMA, fp32 accumulate)",
            "roofline": {"bound": "mfma", "kernel": "agp_netvlad::netvlad_mfma_kernel<256> (v_mfma_f32_32x32x2_f32) + netvlad_finish_kernel",
                         "achieved": round(tf, 2), "peak": PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_F32_MATRIX_TFLOPS, 4),
                         "algorithmic_gflop_per_batch": round(flop / 1e9, 3),
                         "hbm": {"algorithmic_bytes_per_batch": nbytes, "achieved_GBps": round(nbytes / (ms * 1e-3) / 1e9, 1),
                                 "frac_of_8TBps": round(nbytes / (ms * 1e-3) / 8e12, 4)}},
            "valu_kernel_ms_per_batch": round(ms_valu, 4), "speedup_over_valu_kernel": round(ms_valu / ms, 2),
            "max_abs_difference_from_valu_kernel": float((y - yv).abs().max())}


def conv_roofline(args, embed, ops, rank, c2):
    """`roofline` of the line: the 3x3 stride-1 convolutions (the dominant kernel family), HIP events on the launch stream around
    every conv launch of ONE single-stream eager pass; `conv_family` = all conv launches of that pass."""
    # ---- roofline of the dominant kernel (implicit-GEMM conv), events on the launch stream
    for _ in range(2):             # the eager passes below run on the default stream: build its workspaces first
        embed(serial=True)
        embed()
    torch.cuda.synchronize()
    ops.CONV_PROFILE = []
    embed(serial=True)             # one stream: a launch's events must bracket only that launch
    torch.cuda.synchronize()
    prof, ops.CONV_PROFILE = ops.CONV_PROFILE, None
    conv_ms = sum(p[0].elapsed_time(p[1]) for p in prof)
    conv_macs = sum(p[2] for p in prof)
    if args.verbose and rank == 0:
        for e0, e1, m, shp in prof:
            ms = e0.elapsed_time(e1)
            print(f"conv n,ho,wo,cin,cout,kh,kw,s={shp} {ms:.4f} ms {2 * m / ms / 1e9:.1f} TFLOP/s", file=sys.stderr)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    embed()
    e1.record()
    torch.cuda.synchronize()
    embed_ms = e0.elapsed_time(e1)
    achieved = 2.0 * conv_macs / (conv_ms * 1e-3) / 1e12
    # the dominant KERNEL FAMILY: the 3x3 stride-1 convs (fblock64 for the 64-channel blocks, igemm_kxrw for the others)
    kxr = [p for p in prof if p[3][5] == 3 and p[3][6] == 3 and p[3][7] == 1]
    kxr_ms = sum(p[0].elapsed_time(p[1]) for p in kxr)
    kxr_macs = sum(p[2] for p in kxr)
    kxr_achieved = 2.0 * kxr_macs / (max(kxr_ms, 1e-9) * 1e-3) / 1e12
    # HBM traffic per launch from the PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    # separate runs, gfx950 correction applied; tools/summarize_profiles.py).  PMC counters cannot be read from
    # inside the process, so the committed summary is quoted.
    traffic = kxr_traffic = None
    pmc_file = f"profiles/{PROFILE_TAG}_pmc_conv_p{args.prec}.json"
    traffic_note = f"HBM bytes per launch ({pmc_file}, separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"
    try:
        with open(os.path.join(ROOT, pmc_file)) as f:
            pmc = json.load(f)
        if pmc.get("csrc_sha16") == bench_inputs.kernel_source_sha16(ROOT) and not c2:
            traffic = round(pmc["conv_hbm_bytes_per_launch"])
            kxr_traffic = round(pmc["conv3x3_family_hbm_bytes_per_launch"])
        else:
            traffic_note = f"null: {pmc_file} was measured on other kernel sources (csrc_sha16 differs) or another workload"
    except Exception:
        traffic_note = f"null: no {pmc_file}"
    passes = {2: 1.5 if args.lo_fp8 else 2, 3: 3, 4: 1}[args.prec]
    roofline = {
        "bound": "mfma", "kernel": "agp_fb::fblock64_kernel (a whole 64-channel BasicBlock = two 3x3 convs per launch, intermediate in LDS) / "
                                   "agp_igemm::igemm_kxrw_kernel (cout % 128 == 0: 256 x 128 tiles) -- every 3x3 stride-1 conv of a step, the query and "
                                   "the database network's work of a layer as one grouped launch" if args.prec == 4 else
                                   "agp_igemm::igemm_kxr_kernel (every 3x3 stride-1 conv of a step; implicit GEMM with horizontal-tap reuse)",
        "achieved": round(kxr_achieved, 2), "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s",
        "frac": round(kxr_achieved / PEAK_BF16_DENSE_TFLOPS, 4), "traffic": kxr_traffic,
        "traffic_unit": traffic_note,
        "launches_per_step": len(kxr), "avg_launch_ms": round(kxr_ms / max(len(kxr), 1), 4),
        "algorithmic_gflop_per_launch": round(2.0 * kxr_macs / max(len(kxr), 1) / 1e9, 3),
        "kernel_ms_per_step": round(kxr_ms, 3), "mfma_passes_per_algorithmic_flop": passes,
        # the whole conv family (stem, 1x1 / stride-2 convs on the generic kernel, 3x3 stride-1 convs)
        "conv_family": {"achieved": round(achieved, 2), "frac": round(achieved / PEAK_BF16_DENSE_TFLOPS, 4), "traffic": traffic,
                        "launches_per_step": len(prof), "avg_launch_ms": round(conv_ms / max(len(prof), 1), 4),
                        "algorithmic_gflop_per_launch": round(2.0 * conv_macs / max(len(prof), 1) / 1e9, 3),
                        "conv_ms_per_step": round(conv_ms, 3)},
        "embed_ms_per_step_eager": round(embed_ms, 3),
    }
    return roofline


def cpu_baseline_measurement(args, opt, modelq, modeldb, data, tiles, b):
    """`cpu_baseline`: the oracle (a PyTorch-CPU port of the reference forward: python-loop fixed-grid ODE, F.conv2d ResNet) timed on
    a bounded sample of the same workload on the host cores.  The ONLY use of oracle/nets in this file."""
    from oracle import nets as onets            # the ONLY use of oracle/ in this file: the timed CPU port
    n = min(b, args.cpu_pairs)
    reps = max(1, args.cpu_pairs // n)
    pq = {k: v.cpu() for k, v in modelq.state_dict().items()}
    pd = {k: v.cpu() for k, v in modeldb.state_dict().items()}
    dc = {k: ([t[:n].cpu() for t in v] if isinstance(v, list) else v[:n].cpu()) for k, v in data.items()}
    tc = tiles[:n].cpu()

    def run(m):
        sub = {k: ([t[:m] for t in v] if isinstance(v, list) else v[:m]) for k, v in dc.items()}
        t0 = time.perf_counter()
        onets.mm_forward_q(sub, pq, opt)
        onets.dbvanilla2d_forward_db({"db_map": tc[:m]}, pd, opt)
        return time.perf_counter() - t0

    with torch.no_grad():
        # PyTorch-CPU scales badly past a few dozen threads on this 2x64-core host (256 threads
        # is >100x slower than 16): pick the best of a short sweep, then time the sample with it.
        best_thr, best_t = None, None
        for thr in (8, 16, 32, 64):
            if thr > (os.cpu_count() or 1):
                continue
            torch.set_num_threads(thr)
            run(1)
            t = run(2)
            if best_t is None or t < best_t:
                best_thr, best_t = thr, t
        torch.set_num_threads(best_thr)
        run(1)
        cdt = sum(run(n) for _ in range(reps))
    return {"value": round(n * reps / cdt, 3), "unit": "pairs/s", "cores": best_thr,
                           "kind": "port", "sample": f"{n * reps} pairs ({reps} passes of {n}) of the same workload (fp32 PyTorch-CPU "
                           "oracle: python-loop fixed-grid ODE, F.conv2d ResNet18), timed after warm-up; "
                           f"thread count chosen from a sweep over 8/16/32/64 (host has {os.cpu_count()} hw threads)"}


def _dev_sync(dev):
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)


def knn_distributed_leg(db, nq_rank, opt, dev, rank, world, parallel, retrieval, reps=3):
    """The N > 1 kNN leg: rank r holds rows shard_range(100000, r, world) of the database and its own
    `nq_rank` queries; `retrieval.distributed_search` = all-gather of the database shards (the north star's exchange step,
    timed: allgather_ms, allgather_GBps = bytes a rank receives / that time) + index planes + search + all-gather of the
    [Q, k] results.  Returns the phase times (median of `reps` calls, MAX over ranks), the index of the last call and the rank's
    queries; checks on every rank that the gathered database equals the unsharded one."""
    dlo, dhi = parallel.shard_range(db.shape[0], rank, world)
    local_db = db[dlo:dhi].to(dev)
    gq = torch.Generator().manual_seed(1000 + rank)
    q = torch.randn(nq_rank, db.shape[1], generator=gq)
    q = (q / q.norm(dim=1, keepdim=True)).to(dev)
    runs = []
    for i in range(reps + 1):
        parallel.barrier()
        tm = {}
        D, I = retrieval.distributed_search(q, local_db, 20, device=dev, prec=opt.knn_precision, timings=tm)
        if i:
            runs.append(tm)
    index = runs[-1]["index"]
    assert tuple(D.shape) == (nq_rank * world, 20) and tuple(I.shape) == (nq_rank * world, 20)
    xb = getattr(index, "_xb", getattr(index, "xb", None))
    gathered_ok = bool(xb is None or torch.equal(xb[:, :db.shape[1]].cpu(), db))

    def med(key):
        v = sorted(r[key] for r in runs)[len(runs) // 2]
        t = torch.tensor([v], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    ag_ms = med("allgather_ms")
    nbytes = runs[-1]["allgather_bytes"]
    return {"database_rows_per_rank": dhi - dlo, "queries_per_rank": nq_rank,
            "allgather_ms": round(ag_ms, 4), "allgather_bytes_received_per_rank": nbytes,
            "allgather_GBps": round(nbytes / (ag_ms * 1e-3) / 1e9, 2) if ag_ms > 0 else None,
            "prepare_index_ms": round(med("prepare_ms"), 4), "first_search_ms": round(med("search_ms"), 4),
            "gather_results_ms": round(med("gather_results_ms"), 4),
            "gathered_database_equals_unsharded": gathered_ok, "index": index, "queries": q}


def knn_measurement(args, opt, dev, rank, world, parallel, retrieval, db_rows=100000, nq_rank=4096, reps=20):
    """BASELINE config C5: exact L2 kNN at DB = 100k x 256, k = 20 (reference test.py:27-32): queries/s of the HIP search (N > 1:
    the database sharded over the ranks and all-gathered, 4096 queries per rank: knn_distributed_leg), the roofline of its coarse kernel, the CPU port timed beside it on a bounded sample, and
    the in-run PARITY check: indices of the GPU search against the CPU port's and against an exact fp64 brute force, and Recall@1/5
    (test.py:73-83) of both on queries with planted positives."""
    g = torch.Generator().manual_seed(1)
    db = torch.randn(db_rows, 256, generator=g)
    db = db / db.norm(dim=1, keepdim=True)
    q = torch.randn(nq_rank, 256, generator=g)
    dist_leg = None
    if world > 1:
        # N > 1 (north star: "all-gather on the eval descriptor database over xGMI"; reference test.py:125-176 fills ONE [N,256]
        # matrix): every rank owns 100000 / world database rows -- what a sharded extraction loop leaves it with -- and 4096
        # queries OF ITS OWN (weak scaling); retrieval.distributed_search all-gathers the database (timed), builds the index and
        # searches the rank's queries; the steady-state rate below then searches that index
        dist_leg = knn_distributed_leg(db, nq_rank, opt, dev, rank, world, parallel, retrieval)
        index, q = dist_leg.pop("index"), dist_leg.pop("queries")
    else:
        db = db.to(dev)
        q = (q / q.norm(dim=1, keepdim=True)).to(dev)
        index = retrieval.IndexFlatL2(256, device=dev, prec=opt.knn_precision)
        index.add(db)
    nq_total = nq_rank * world
    for _ in range(3):
        index.search_device(q, 20)
    # a search is ~0.6 ms: time 5 blocks of 20 back-to-back searches and report the median block
    # (one host hiccup inside a 3 ms window used to move this number by 2-10x)
    def timed_blocks(qq):
        blocks = []
        for _ in range(5):
            parallel.barrier()
            _dev_sync(dev)
            t0 = time.perf_counter()
            for _ in range(reps):
                index.search_device(qq, 20)
            _dev_sync(dev)
            parallel.barrier()
            blocks.append(time.perf_counter() - t0)
        t = sorted(blocks)[len(blocks) // 2]
        if world > 1:
            tt = torch.tensor([t], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t = float(tt.item())
        return t
    kdt = timed_blocks(q)
    res = {"metric": "kNN queries/sec @ DB=100k x 256, k=20, exact L2", "value": round(nq_total * reps / kdt, 1),
                  "unit": "queries/s", "nq": nq_total, "algorithmic_mflop_per_query": 51.2,
                  "achieved_tflops": round(nq_total * reps * 51.2e6 / kdt / 1e12, 2)}
    if dist_leg is not None:
        # the strong-scaled figure beside it: the single-GPU leg's 4096 queries split over the ranks (4096 / world per search call)
        slo, shi = parallel.shard_range(nq_rank, rank, world)
        sdt = timed_blocks(q[: shi - slo])
        dist_leg["strong_scaled_queries_per_s"] = round(nq_rank * reps / sdt, 1)
        dist_leg["strong_scaled_queries_per_rank"] = shi - slo
        res["scaling"] = "weak (4096 queries per rank; database sharded 100000 / world rows per rank and all-gathered)"
        res["distributed"] = dist_leg
    if dev.type != "cuda":
        return res               # (the control flow under gloo on CPU: tests/test_parallel_gloo.py)
    # roofline of its dominant kernel (the fp16 coarse distance pass: 2 N D flop per query): HIP events on the launch stream
    # around the search's first stage alone (agp_knn_coarse_pass: query preparation + coarse pass)
    for _ in range(3):
        index.coarse_pass_device(q)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        index.coarse_pass_device(q)
    e1.record()
    torch.cuda.synchronize()
    coarse_ms = e0.elapsed_time(e1) / reps
    nq_local = q.shape[0]
    ktf = nq_local * 51.2e6 / (coarse_ms * 1e-3) / 1e12
    res["roofline"] = {
        "bound": "mfma", "kernel": "agp_knn::coarse_f16_w4_kernel<256> (fp16 coarse distances; four waves per workgroup, one per SIMD, "
                                   "64 queries resident in each wave's registers, the query preparation fused into its prologue)",
        "achieved": round(ktf, 2), "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s", "frac": round(ktf / PEAK_BF16_DENSE_TFLOPS, 4),
        "avg_launch_ms": round(coarse_ms, 4), "algorithmic_gflop_per_launch": round(nq_local * 51.2e-3, 2),
        "search_ms": round(kdt / reps * 1e3, 4), "traffic": None}
    # the same index at 16384 queries per search call (a test set's worth; the headline figure above keeps round 3's 4096): the
    # fixed costs of a search -- query fragments into registers, the tail of the last database tiles -- are spread over 4x the work
    if world == 1:
        qb = torch.randn(16384, 256, generator=g)
        qb = (qb / qb.norm(dim=1, keepdim=True)).to(dev)
        for _ in range(3):
            index.search_device(qb, 20)
        bl = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                index.search_device(qb, 20)
            torch.cuda.synchronize()
            bl.append(time.perf_counter() - t0)
        res["queries_per_s_at_16384_per_search"] = round(16384 * 5 / sorted(bl)[len(bl) // 2], 1)
    # HBM bytes per coarse launch from the PMC passes of tools/knn_bench.py on the same problem (profiles/<tag>_pmc_knn.json;
    # quoted only while the kernel sources still hash to the value it was measured at)
    try:
        knn_pmc = f"profiles/{PROFILE_TAG}_pmc_knn.json"
        with open(os.path.join(ROOT, knn_pmc)) as f:
            kp = json.load(f)
        if kp.get("csrc_sha16") == bench_inputs.kernel_source_sha16(ROOT) and world == 1:
            kk = "coarse_f16_w4_kernel" if "coarse_f16_w4_kernel" in kp["kernels"] else "coarse_f16_kernel"
            res["roofline"]["traffic"] = round(kp["kernels"][kk]["hbm_mb_per_launch"] * 1e6)
            res["roofline"]["traffic_unit"] = (f"HBM bytes per coarse launch ({knn_pmc}: separate rocprofv3 --pmc "
                                                      "FETCH_SIZE / WRITE_SIZE passes of tools/knn_bench.py, 4096 queries)")
            res["roofline"]["mfma_busy_frac"] = round(kp["kernels"][kk].get("mfma_busy_frac", 0.0), 3)
        elif world > 1:
            res["roofline"]["traffic_unit"] = f"null: {knn_pmc} is a single-GPU measurement (quoted at N = 1 only)"
        else:
            res["roofline"]["traffic_unit"] = f"null: {knn_pmc} was measured on other kernel sources (csrc_sha16 differs)"
    except Exception:
        res["roofline"]["traffic_unit"] = f"null: no profiles/{PROFILE_TAG}_pmc_knn.json"
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import knn as oknn        # checker used as the timed CPU port (faiss's BLAS path restated in numpy fp32)
        qs = q[:args.cpu_knn_queries].cpu().numpy()
        dbh = db.cpu().numpy()
        oknn.knn_l2_faisslike_fp32(qs[:32], dbh, 20)
        t0 = time.perf_counter()
        oknn.knn_l2_faisslike_fp32(qs, dbh, 20)
        cdt = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": round(qs.shape[0] / cdt, 1), "unit": "queries/s", "cores": os.cpu_count(),
                                      "kind": "port", "sample": f"{qs.shape[0]} of the same queries against the same 100k x 256 "
                                      "database: numpy fp32 sgemm expansion (||x||^2 + ||y||^2 - 2<x,y>, blocks of queries) + "
                                      "argpartition(k) + sort of the k kept -- the shape of faiss IndexFlatL2's BLAS path (sgemm + "
                                      "top-k selection) in numpy, NOT faiss itself (its heap selection is fused and threaded); "
                                      "numpy's BLAS threads = all host cores"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # `timed`: the very searches behind the figures above (4096 and 16384 queries per call run coarse_f16_w4_kernel; the
        # sampled 512-query search below runs coarse_f16_kernel<D, 2>) -- a sample of THEIR rows against the fp64 brute force
        res["parity"] = knn_parity(index, db, q[:args.cpu_knn_queries], dev, timed=[q, qb])
        pr = res["parity"]
        if not (pr["indices_equal_fp64_bruteforce_64q"] and pr["recall_at_1_5_equal"] and pr["timed_searches_equal_fp64_bruteforce"]):
            # loud, not only a field of the JSON line: this is how round 5's LDS ring-slot race was found (profiles/README.md);
            # a throughput figure of a search that returns wrong neighbours is not a measurement: it is withdrawn and bench.py
            # exits non-zero (bench.py reads `parity_failed`)
            print("bench.py: kNN PARITY FAILED -- GPU indices differ from the fp64 brute force / the CPU port: " + json.dumps(pr),
                  file=sys.stderr, flush=True)
            res["value_withdrawn"] = res["value"]
            res["value"] = None
            res["parity_failed"] = True
    return res


def knn_parity(index, db, q, dev, timed=()):
    """Outside every timed region.  `timed`: the query sets of the timed searches; for each, 128 of its rows (the first 64 and 64
    spread over the set: every query block of the four-wave coarse kernel's grid is sampled) against the fp64 brute force.  (i) the sampled bench queries: GPU indices == the CPU port's (numpy fp32 sgemm expansion, the
    checker oracle/knn.py) and == an exact fp64 brute force on the first 64; (ii) queries with PLANTED positives (database row +
    N(0, 0.05^2) noise): Recall@1 / Recall@5 by the arithmetic of test.py:73-83 from the GPU's and from the CPU
    port's predictions."""
    import numpy as np
    from oracle import knn as oknn
    dbh = db.cpu().numpy()
    qs = q.cpu().numpy()
    _, Ig = index.search_device(q, 20)
    Ig = Ig.cpu().numpy()
    _, Ic = oknn.knn_l2_faisslike_fp32(qs, dbh, 20)[:2]
    _, I64, _ = oknn.knn_l2_fp64(qs[:64], dbh, 20)
    g = torch.Generator().manual_seed(7)
    planted = torch.randint(0, dbh.shape[0], (qs.shape[0],), generator=g)
    qp = db[planted.to(dev)] + 0.05 * torch.randn(qs.shape[0], dbh.shape[1], generator=g).to(dev)
    qp = qp / qp.norm(dim=1, keepdim=True)
    _, Pg = index.search_device(qp, 5)
    Pg = Pg.cpu().numpy()
    Pc = oknn.knn_l2_faisslike_fp32(qp.cpu().numpy(), dbh, 5)[1]
    tgt = planted.numpy()[:, None]

    def recall(P, n):
        return float(np.mean(np.any(P[:, :n] == tgt, axis=1)) * 100)
    rg, rc = [recall(Pg, 1), recall(Pg, 5)], [recall(Pc, 1), recall(Pc, 5)]
    timed_rows, timed_ok = [], True
    for qq in timed:
        nq = int(qq.shape[0])
        rows = np.unique(np.concatenate([np.arange(min(64, nq)), np.linspace(0, nq - 1, 64).astype(np.int64)]))
        _, It = index.search_device(qq, 20)              # the timed call itself (same query count -> same kernel)
        It = It.cpu().numpy()[rows]
        _, Ir, _ = oknn.knn_l2_fp64(qq[torch.from_numpy(rows).to(qq.device)].cpu().numpy(), dbh, 20)
        bad = int(np.any(It != Ir, axis=1).sum())
        timed_rows.append({"queries_per_search": nq, "rows_checked": int(rows.size), "rows_differing": bad})
        timed_ok = timed_ok and bad == 0
    return {"queries": int(qs.shape[0]), "timed_searches_equal_fp64_bruteforce": timed_ok, "timed_searches": timed_rows, "indices_equal_cpu_port": bool(np.array_equal(Ig, Ic)),
            "rows_differing_from_cpu_port": int(np.any(Ig != Ic, axis=1).sum()),
            "indices_equal_fp64_bruteforce_64q": bool(np.array_equal(Ig[:64], I64)),
            "recall_at_1_5_gpu": rg, "recall_at_1_5_cpu_port": rc, "recall_at_1_5_equal": rg == rc,
            "planted": "database row + N(0, 0.05^2) noise, renormalised; 100k x 256, k = 20 / 5"}
