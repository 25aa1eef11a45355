"""Host -> device input staging.

The reference moves every tensor of a batch to the GPU synchronously at the top of the step
(`data_dict[k] = v.to(args.device)`, train.py:303-304) after its DataLoader workers have decoded, resized, normalised
(fp32) and width-concatenated the camera tiles on the CPU (datasets/datasets_ws_nuscenes.py:604-634).  Here the host
ships the decoded uint8 tiles (4x fewer bytes than the normalised fp32 panorama) and the device does the rest
(ops.pack_cameras_u8 -> agp_pack_u8_cams_to_nhwc); this module is the transport:

    PinnedRing   `depth` slots, each a pinned host buffer + a device buffer per named tensor.  `upload(slot)` enqueues
                 the slot's H2D copies on a dedicated COPY stream (after the consumer of the slot's previous contents
                 has released it); `acquire(slot)` makes the caller's stream wait for that upload and returns the device
                 tensors; `release(slot)` marks them consumed.  With depth >= 2 the upload of batch i+1 runs under the
                 compute of batch i: PCIe never sits in the step's critical path as long as a batch's bytes move faster
                 than the step computes (57.8 MB of uint8 tiles per 64 panoramas: 0.9 ms at PCIe 5 x16 against a 2.4 ms step).

Measured on the MI355X box (round 2, bench.py --h2d, 67.4 MB of uint8 tiles per 64-pair step): resident inputs 2.42 ms per
step, this ring 2.8 - 2.9 ms.  The copy itself is not the bound (hipMemcpyAsync of 64 MB: 1.24 ms = 54 GB/s, and ring depth
2, 3, 4 give the same step); running it beside the step costs: every kernel of the step gets ~4.5 % slower (rocprofv3
kernel trace with / without the copy) and the rest is launch latency (AQL packets and completion signals cross the same
PCIe link the bulk read saturates).  Tried and rejected: splitting the upload into 8 copies (3.3 ms), blit copies
(HSA_ENABLE_SDMA=0: 3.7 ms), a bounded-grid copy kernel reading the pinned arena (8 workgroups: 2.75 ms, more are worse),
packing straight from the pinned arena (zero-copy: 3.9 ms -- PCIe-stalled waves take the convolutions' wave slots).

Loader workers write straight into `ring.host(slot)[name]` (pinned, page-locked: `torch.from_numpy` views of it can be
handed to worker processes through shared memory); nothing here touches pixel values.
"""
import torch


class PinnedRing:
    def __init__(self, spec, depth=2, device="cuda"):
        """spec: {name: (shape, dtype)} of one batch."""
        if depth < 1:
            raise ValueError("PinnedRing: depth >= 1")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("PinnedRing stages into GPU memory (agplace_amd has no CPU path)")
        self.depth = depth
        # ONE pinned arena and ONE device arena per slot, the named tensors are views into them: a slot uploads as a
        # single copy (measured on the MI355X box: a 64 MB pinned copy moves at 35 GB/s, an 8 MB one at 11 GB/s --
        # two copies per step, 58 MB + 10 MB, were slower than the step they should hide under)
        offs, total = {}, 0
        for k, (shape, dt) in spec.items():
            nbytes = int(torch.Size(shape).numel()) * torch.empty((), dtype=dt).element_size()
            offs[k] = (total, nbytes, tuple(shape), dt)
            total = (total + nbytes + 255) // 256 * 256
        self._harena = [torch.empty(total, dtype=torch.uint8).pin_memory() for _ in range(depth)]
        self._darena = [torch.empty(total, dtype=torch.uint8, device=self.device) for _ in range(depth)]

        def views(arena):
            return {k: arena[o:o + n].view(dt).view(shape) for k, (o, n, shape, dt) in offs.items()}
        self._host = [views(a) for a in self._harena]
        self._dev = [views(a) for a in self._darena]
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._ready = [torch.cuda.Event() for _ in range(depth)]      # upload of the slot has finished
        self._free = [torch.cuda.Event() for _ in range(depth)]       # the consumer has finished with the slot's device tensors
        self._free_valid = [False] * depth
        self.bytes_per_batch = total

    def host(self, slot):
        """The slot's pinned host tensors: the producer (dataloader) fills them in place."""
        return self._host[slot % self.depth]

    def device_tensors(self, slot):
        """The slot's device tensors (static addresses for the life of the ring: a captured hipGraph may bake them in).
        Use `acquire` to also order a stream behind the slot's upload."""
        return self._dev[slot % self.depth]

    def host_arena(self, slot):
        """The slot's whole pinned buffer / device buffer as flat uint8 tensors (one copy moves a slot)."""
        return self._harena[slot % self.depth]

    def device_arena(self, slot):
        return self._darena[slot % self.depth]

    def upload(self, slot):
        """Enqueue host -> device copies of the slot on the copy stream (asynchronous; returns at once)."""
        s = slot % self.depth
        with torch.cuda.stream(self.copy_stream):
            if self._free_valid[s]:
                self.copy_stream.wait_event(self._free[s])          # do not overwrite tensors a kernel may still read
            self._darena[s].copy_(self._harena[s], non_blocking=True)
            self._ready[s].record(self.copy_stream)

    def acquire(self, slot, stream=None):
        """Device tensors of the slot, valid on `stream` (default: the current stream) once its upload has landed."""
        s = slot % self.depth
        (stream or torch.cuda.current_stream(self.device)).wait_event(self._ready[s])
        return self._dev[s]

    def release(self, slot, stream=None):
        """The work enqueued so far on `stream` is the last reader of the slot's device tensors."""
        s = slot % self.depth
        self._free[s].record(stream or torch.cuda.current_stream(self.device))
        self._free_valid[s] = True
