"""Picture-granular sharding of the hot path over the GPUs of one node: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo"
in the CPU tests).

What shards: pictures.  MCTF-filtered pictures and frame-/GOP-parallel pictures are independent units of work once their inputs are on the device; CTUs of
one picture are not (WPP + CABAC state), so nothing finer than a picture crosses a device boundary and no collective sits in the per-candidate data path
("scaling": "weak").  The one real exchange step is picture-granular: the rank that owns a picture other ranks depend on — an original picture inside another rank's MCTF
window, a reconstructed picture that is a reference of pictures encoded elsewhere — publishes it to every rank.  `PictureExchange` is that step: a decoded-picture-buffer
ring replicated on every rank, filled by broadcasts that run on their own stream so that the transfer of picture p+1 overlaps the work on picture p.

(Inside ONE encoder process the same mapping is done by the binding with device-to-device copies, bindings/vvenc: `$VVHIP_GPUS`.)
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None):
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # (dmabuf IPC: what RCCL's buffer sharing needs on this pool)
        if backend is None:
            # $VVHIP_DIST_BACKEND=gloo + $VVHIP_SHARE_DEVICE=1: every rank on GPU 0, collectives through gloo — how the N>1 path is exercised on a 1-GPU box (tests)
            backend = os.environ.get("VVHIP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if os.environ.get("VVHIP_SHARE_DEVICE") == "1":
            local_rank = 0
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def frames_of_rank(n_frames, rank, world):
    """round-robin picture ownership (picture p -> rank p % world)"""
    return list(range(rank, n_frames, world))


def owner_of(frame, world):
    return frame % world


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def broadcast_picture(storage, src_rank, async_op=False):
    """broadcast one picture plane (int16 tensor incl. margins) from its owner to every rank; RCCL over xGMI on GPUs.
    1080p luma incl. margin = 5.2 MB, 4K = 18.6 MB: one message per picture, never per block."""
    if dist.is_initialized():
        # neither RCCL nor gloo has a 16-bit integer type: ship the plane as bytes (same memory, no copy)
        return dist.broadcast(storage.view(torch.uint8), src=src_rank, async_op=async_op)
    return None


class PictureExchange:
    """Decoded-picture-buffer ring replicated on every rank.

    `slots` pictures of `planes` tensors each (luma + the two chroma planes, with their margins — what a reference picture is).  The owner of picture p
    writes it into slot(p) and every rank calls publish(p, owner): one broadcast per picture (its planes are views into one allocation), issued on the exchange's own stream (GPU) so it overlaps the kernels of the
    picture being worked on; wait(p) makes the compute stream wait for it (stream-ordered, no host sync).  Every rank must publish the same pictures in the same order.
    bytes_published counts what this rank sent or received."""

    def __init__(self, plane_shapes, slots=2, device="cpu", dtype=torch.int16):
        self.device = torch.device(device)
        # one contiguous allocation per picture: the planes are views into it, the whole picture travels in ONE collective
        sizes = [int(torch.Size(s).numel()) for s in plane_shapes]
        self.flat = [torch.zeros(sum(sizes), dtype=dtype, device=self.device) for _ in range(slots)]
        self.slots = []
        for f in self.flat:
            off, planes = 0, []
            for s, n in zip(plane_shapes, sizes):
                planes.append(f[off:off + n].view(s))
                off += n
            self.slots.append(planes)
        self.n_slots = slots
        self.pending = {}          # frame -> [work handles] / event
        self.bytes_published = 0
        self.is_cuda = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(device=self.device) if self.is_cuda else None

    def slot(self, frame):
        return self.slots[frame % self.n_slots]

    def publish(self, frame, owner, readers=(), after=None):
        """readers: extra (compute) streams whose earlier work reads / writes this slot — the broadcast waits for them too.
        after(planes): device work derived from the received picture (e.g. its tiled copy), enqueued on the exchange stream behind the broadcast and covered by wait()"""
        planes = self.slot(frame)
        self.bytes_published += sum(p.numel() * p.element_size() for p in planes)
        if not dist.is_initialized():
            if after is not None:
                after(planes)
            self.pending[frame] = None
            return
        if self.is_cuda:
            # the broadcast must see what the compute stream wrote into the slot (owner) / must not overwrite a slot still being read (others)
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
            for r in readers:
                self.stream.wait_stream(r)
            with torch.cuda.stream(self.stream):
                broadcast_picture(self.flat[frame % self.n_slots], owner)
                if after is not None:
                    after(planes)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            self.pending[frame] = ev
        else:
            self.pending[frame] = [broadcast_picture(self.flat[frame % self.n_slots], owner, async_op=True)]
            if after is not None:
                for w in self.pending[frame]:
                    w.wait()
                self.pending[frame] = []
                after(planes)

    def wait(self, frame, streams=None):
        """streams: the compute streams that will read the picture (default: the current stream)"""
        h = self.pending.pop(frame, None)
        if h is None:
            return self.slot(frame)
        if self.is_cuda:
            for st in (streams or [torch.cuda.current_stream(self.device)]):
                st.wait_event(h)
        else:
            for w in h:
                w.wait()
        return self.slot(frame)


def run_sharded_gops(n_frames, gop, rank, world, exchange, produce_key, process_dependent):
    """The dependency pattern of a sharded random-access sequence, as a driver (the CPU tests run it over gloo, bench.py's multi-GPU step mirrors it):
    key pictures 0, gop, 2*gop, .. are owned round-robin (key k*gop -> rank k % world); the owner `produce_key(p, slot)`s the picture (fills the slot: its
    reconstruction / the original it read) and every rank publishes it.  The pictures between two keys depend on those two keys only, so they are dealt round-robin to
    the ranks and each rank runs `process_dependent(q, prev_key_slot, next_key_slot)` for its own — the broadcast of the NEXT key is already in flight while they run.
    Needs exchange.n_slots >= 3.  Returns {picture: result} for the dependent pictures this rank processed."""
    assert exchange.n_slots >= 3
    keys = list(range(0, n_frames, gop))
    results = {}

    def publish(i):
        k = keys[i]
        if i % world == rank:
            produce_key(k, exchange.slot(i))
        exchange.publish(i, i % world)          # the ring is indexed by key number

    for i in range(min(2, len(keys))):
        publish(i)
    dealt = 0
    for i in range(len(keys) - 1):
        if i + 2 < len(keys):
            publish(i + 2)                       # in flight while the pictures between key i and key i + 1 are processed
        prev_slot, next_slot = exchange.wait(i), exchange.wait(i + 1)
        for q in range(keys[i] + 1, keys[i + 1]):
            if dealt % world == rank:
                results[q] = process_dependent(q, prev_slot, next_slot)
            dealt += 1
    return results
