"""Query network and database network embedded TOGETHER (inference).

The reference calls the two models one after the other (`modelq(data_dict, 'q')`, `model(data_dict, 'db')`:
train.py:308,316; test.py:128,161; datasets_ws_nuscenes.py:1220-1227).  Both are ResNet trunks of one architecture
by default (tools/options.py:85-104), so each layer's convolution exists twice per step with different weights and
very different sizes: a 6-camera panorama is six aerial tiles' worth of pixels.  On MI355X the small launches are the
expensive ones (a 3x3 conv over 64 aerial tiles fills a third of the chip for one wave of workgroups), so
`embed_pair` advances the two trunks in lock-step and issues every layer as ONE grouped launch
(resnet.forward_maps_multi -> agp_conv2d_fwd_grouped); everything behind the trunks is the models' own forward.
Outputs are bit-identical to calling the two models separately (tests/test_gpu_models.py).
"""
import torch


from . import resnet

RIDER = True      # False: the database head as a launch of its own behind the query network's tail


def can_pair(modelq, modeldb, qdata, dbdata):
    if modelq.training or modeldb.training or torch.is_grad_enabled():
        return False
    fq, fdbs = modelq.image_fe.fe, [m.fe for m in modeldb.dbimage_fes]
    db_map = dbdata['db_map']
    nmap = db_map.shape[-4]                     # [..., nmap, 3, h, w] and uint8 [..., nmap, h, w, 3] alike
    if modelq.opt.mfma_precision != modeldb.opt.mfma_precision or db_map.dim() not in (5, 6):
        return False
    if modeldb.opt.share_dbfe is True and nmap > 1:
        return False
    return all((f.fe_type, f.nstages) == (fq.fe_type, fq.nstages) for f in fdbs[:nmap])


def embed_pair(modelq, modeldb, qdata, dbdata):
    """(modelq(qdata, 'q'), modeldb(dbdata, 'db')) with the image trunks run in lock-step.  Falls back to the two
    separate forwards when the models cannot be paired (training, different trunk architectures)."""
    if not can_pair(modelq, modeldb, qdata, dbdata):
        return modelq(qdata, mode='q'), modeldb(dbdata, mode='db')
    k = modelq.opt.query_substreams
    img = qdata['query_image']
    db_map = dbdata['db_map']
    b = img.shape[0]
    if k > 1 and 'coords' not in qdata and b % k == 0 and b >= 2 * k and db_map.shape[0] % k == 0:
        return _embed_pair_substreams(modelq, modeldb, qdata, dbdata, k)
    return _embed_pair_one(modelq, modeldb, qdata, dbdata)


def _embed_pair_one(modelq, modeldb, qdata, dbdata, q_rows=None, db_rows=None):
    opt = modelq.opt
    prec = opt.mfma_precision
    image = modelq.query_image(qdata)
    db_map = dbdata['db_map']
    u8 = db_map.dtype == torch.uint8            # decoded tiles [b,(ndb,)nmap,h,w,3]: normalised on the device (DBVanilla2D.forward_db)
    if db_map.dim() == 5:
        db_map = db_map.unsqueeze(1)
    nets, xs, lms, fps = [modelq.image_fe.fe], [image], [[]], [modelq.final_pool_request()]
    nmap = db_map.shape[2]
    for i in range(nmap):
        nets.append(modeldb.dbimage_fes[i].fe)
        if u8:
            bb, ndb, _, h, w, _ = db_map.shape
            xs.append(db_map[:, :, i].reshape(bb * ndb, 1, h, w, 3))
        else:
            bb, ndb, _, c, h, w = db_map.shape
            xs.append(db_map[:, :, i].reshape(bb * ndb, c, h, w))
        lms.append(None)
        fps.append(modeldb.final_pool_request(i))
    maps = resnet.forward_maps_multi(nets, xs, prec=prec, level_means=lms, final_pools=fps)
    # the database network's head (one small vector program) rides in the launch of the query network's first program instead of
    # following the query network's tail on the stream: both are latency chains on a handful of CUs
    head = [] if RIDER else None
    out_db = modeldb.forward_db(dbdata, trunk_maps={i: (maps[1 + i], fps[1 + i]) for i in range(nmap)}, out_rows=db_rows,
                                defer_head=head)
    out_q = modelq.forward_q(qdata, image_maps=(maps[0], lms[0], fps[0]), out_rows=q_rows, rider=head)
    if head:
        head.pop().run()                    # not taken along (per-op query path, or the two programs do not fit one launch)
    return out_q, out_db


def _slice(d, b, lo, hi):
    out = {}
    for name, v in d.items():
        if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == b:
            out[name] = v[lo:hi]
        elif isinstance(v, (list, tuple)) and all(torch.is_tensor(t) and t.shape[0] == b for t in v):
            out[name] = [t[lo:hi] for t in v]
        else:
            out[name] = v
    return out


def _embed_pair_substreams(modelq, modeldb, qdata, dbdata, k):
    """The pairs as k sub-batches on k HIP streams (the caller's + k-1 owned by the query module, as in
    MM._forward_q_substreams): one sub-batch's fusion tail (a latency-bound chain of small launches) runs under the
    other sub-batches' convolutions.  Same arithmetic per sample."""
    dev = qdata['query_image'].device
    cur = torch.cuda.current_stream(dev)
    key = (str(dev), k, cur.cuda_stream)
    pool = modelq.__dict__.setdefault('_substream_pool', {})
    if key not in pool:
        pool[key] = [torch.cuda.Stream(device=dev) for _ in range(k - 1)]
    streams = pool[key]
    bq, bd = qdata['query_image'].shape[0], dbdata['db_map'].shape[0]
    hq, hd = bq // k, bd // k
    outs = [None] * k
    # whole-batch outputs: every sub-batch writes its row slice, nothing is concatenated afterwards
    D = modelq.opt.mm_stg2fuse_dim
    fq = {name: torch.empty((bq, D), dtype=torch.float32, device=dev) for name in modelq.OUT_KEYS}
    dbm = dbdata['db_map']
    ndb = dbm.shape[1] if dbm.dim() == 6 else 1
    fd = torch.empty((bd * ndb, 256), dtype=torch.float32, device=dev)

    def one(i):
        return _embed_pair_one(modelq, modeldb, _slice(qdata, bq, i * hq, (i + 1) * hq), _slice(dbdata, bd, i * hd, (i + 1) * hd),
                               q_rows={name: t[i * hq:(i + 1) * hq] for name, t in fq.items()},
                               db_rows=fd[i * hd * ndb:(i + 1) * hd * ndb])
    for i, st in enumerate(streams):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            outs[i + 1] = one(i + 1)
    outs[0] = one(0)
    for i, st in enumerate(streams):
        cur.wait_stream(st)
        for o in outs[i + 1]:
            for t in o.values():
                t.record_stream(cur)
    for t in list(fq.values()) + [fd]:
        for st in streams:
            t.record_stream(st)
    from . import ops
    out_q = {name: ops.join_rows([o[0][name] for o in outs]) for name in outs[0][0]}
    out_db = {name: ops.join_rows([o[1][name] for o in outs]) for name in outs[0][1]}
    return out_q, out_db


class CapturedPair:
    """`embed_pair` captured into a hipGraph on a stream of its own (static input tensors: refill them in place between replays).

    replay() enqueues one replay and -- every `poll_every` replays -- looks at the query model's voxel-range mirror in pinned host
    memory (MM.poll_voxel_range: no stream is synchronised) and raises ValueError when an earlier replay embedded a cloud outside
    the device-side coordinate manager's limits; finish() waits for the stream and checks the replays the polls could not have
    seen yet.  Without `coords` in qdata the polls are no-ops."""

    def __init__(self, modelq, modeldb, qdata, dbdata, stream=None, warmup=2, poll_every=8):
        dev = qdata['query_image'].device
        self.modelq, self.modeldb, self.qdata, self.dbdata = modelq, modeldb, qdata, dbdata
        self.stream = stream if stream is not None else torch.cuda.Stream(device=dev)
        self.poll_every, self._n = max(1, int(poll_every)), 0
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):             # workspaces and weight planes are built outside the capture
                embed_pair(modelq, modeldb, qdata, dbdata)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="thread_local"):
            self.out_q, self.out_db = embed_pair(modelq, modeldb, qdata, dbdata)

    def replay(self):
        self._n += 1
        if self._n % self.poll_every == 0:
            self.modelq.poll_voxel_range()
        # what the caller's stream has enqueued so far (the refill of the static inputs) comes first
        self.stream.wait_stream(torch.cuda.current_stream(self.stream.device))
        with torch.cuda.stream(self.stream):
            self.graph.replay()
        return self.out_q, self.out_db

    def finish(self):
        self.stream.synchronize()
        if not self.modelq.voxel_coords_in_range():
            self.modelq._raise_voxel_range("a replayed")
        return self.out_q, self.out_db
