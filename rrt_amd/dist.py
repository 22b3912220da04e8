"""Row-tile sharding of one frame over the ranks of a torch.distributed job, and the final gather.

One process per GPU.  The frame is cut into tiles of `tile_rows` rows; tile t belongs to rank
t mod world (interleaving evens out the sky/ground cost gradient of the reference's scenes,
SURVEY.md App. A).  There is no data-path communication while rendering; the only exchange is
ONE gather of the per-rank row blocks to rank 0 (backend "nccl" = RCCL over xGMI on MI355X,
"gloo" in the CPU tests), followed by an index_copy that puts rows where they belong.
The RNG is keyed by the global pixel index, so the assembled frame is bit-identical for any
(world, tile_rows).
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_rows(image_height, rank, world, tile_rows):
    """Global row numbers owned by `rank` (ascending) - the same rule as rrtx_shard_rows()."""
    rows = np.arange(image_height, dtype=np.int64)
    return rows[(rows // max(1, tile_rows)) % max(1, world) == rank]


_INDEX_CACHE = {}


def _row_index(image_height, rank, world, tile_rows, device):
    """shard_rows() as a device tensor, built once (an H2D copy per step would serialise the stream)."""
    key = (image_height, rank, world, tile_rows, str(device))
    if key not in _INDEX_CACHE:
        _INDEX_CACHE[key] = torch.from_numpy(shard_rows(image_height, rank, world, tile_rows)).to(device)
    return _INDEX_CACHE[key]


def gather_frame(local_rows, image_height, tile_rows=4, group=None, dst=0):
    """Collects every rank's compact row block on `dst` and returns the assembled frame there.

    local_rows: tensor [n_local_rows, width, 3] holding this rank's rows in shard_rows() order.
    Returns a tensor [image_height, width, 3] on rank `dst`, None elsewhere.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    width = local_rows.shape[1]
    counts = [len(shard_rows(image_height, r, world, tile_rows)) for r in range(world)]
    assert local_rows.shape[0] == counts[rank], "row block does not match this rank's shard"
    if world == 1:
        return local_rows
    most = max(counts)
    # equal-size messages: pad the short blocks (at most tile_rows rows of padding)
    send = local_rows
    if counts[rank] < most:
        send = torch.zeros((most, width, 3), dtype=local_rows.dtype, device=local_rows.device)
        send[: counts[rank]] = local_rows
    send = send.contiguous()
    # gloo (CPU rehearsals and tests) has no device gather: stage through host memory there.  With
    # nccl (= RCCL on ROCm) the blocks go device to device over xGMI.
    staged = dist.get_backend(group) == "gloo" and send.is_cuda
    home = send.device
    if staged:
        send = send.cpu()
    if rank == dst:
        parts = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, gather_list=parts, dst=dst, group=group)
        frame = torch.empty((image_height, width, 3), dtype=local_rows.dtype, device=home)
        for r in range(world):
            frame.index_copy_(0, _row_index(image_height, r, world, tile_rows, home), parts[r][: counts[r]].to(home))
        return frame
    dist.gather(send, gather_list=None, dst=dst, group=group)
    return None


class ShardedRenderer:
    """Rrt bound to this rank's shard, rendering into a torch CUDA tensor (no host round trip)."""

    def __init__(self, scene_file, image_width, image_height, samples_per_pixel, max_depth=50, *, fp64=False, tile_rows=4, sample_chunk=0, seed=1984, group=None,
                 device=None, collect_stats=True, use_bvh=False):
        from .render import Rrt, Scene

        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.h, self.w, self.spp = image_height, image_width, samples_per_pixel
        self.tile_rows = tile_rows
        self.scene = Scene(scene_file, image_width, image_height, fp64=fp64)
        self.rrt = Rrt(image_width, image_height, samples_per_pixel, max_depth, use_bvh=use_bvh, fp64=fp64, device=self.device.index or 0, seed=seed, sample_chunk=sample_chunk,
                       shard_rank=self.rank, shard_count=self.world, tile_rows=tile_rows, collect_stats=collect_stats)
        self.rrt.set_scene(self.scene)
        self.rows = self.rrt.shard_rows()
        assert np.array_equal(self.rows, shard_rows(image_height, self.rank, self.world, tile_rows))
        self.local = torch.zeros((len(self.rows), image_width, 3), dtype=torch.float64 if fp64 else torch.float32, device=self.device)

    def render_local(self):
        """Enqueue one render of this rank's rows on torch's current stream."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.rrt.render_device(self.local.data_ptr(), stream)
        return self.local

    def render(self, dst=0):
        """One full step: render the shard, gather to `dst`.  Returns the frame on dst, None elsewhere."""
        self.render_local()
        if self.world == 1:
            return self.local
        return gather_frame(self.local, self.h, self.tile_rows, self.group, dst)


def main(argv=None):
    """Multi-GPU front end with the reference's flags (main.cpp:69-119): one process per GPU,

        python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
            -m rrt_amd.dist -i scenes/final.txt -o final.png -w 3840 -h 2160 -s 1000

    renders BASELINE.json's 8-GPU configuration: every rank renders its row tiles, one gather over RCCL
    brings them to rank 0, which quantises and writes the PNG (or the PPM to stdout).  Without a launcher
    it runs on one GPU.  -b selects the list scan, -T the tile height (default 4); RRTX_DIST_BACKEND=gloo
    stages the gather through the host (tests, boxes where ranks share a GPU)."""
    import argparse
    import os
    import sys

    ap = argparse.ArgumentParser(prog="python -m rrt_amd.dist", add_help=False)
    ap.add_argument("-i", dest="scene", required=True)
    ap.add_argument("-o", dest="png", default=None)
    ap.add_argument("-w", dest="w", type=int, default=1200)
    ap.add_argument("-h", dest="h", type=int, default=800)
    ap.add_argument("-s", dest="spp", type=int, default=10)
    ap.add_argument("-d", dest="depth", type=int, default=50)
    ap.add_argument("-b", dest="no_bvh", action="store_true")
    ap.add_argument("-T", dest="tile_rows", type=int, default=4)
    ap.add_argument("-S", dest="seed", type=int, default=1984)
    ap.add_argument("--fp64", action="store_true", help="rrtd: double precision")
    ap.add_argument("--help", action="help")
    a = ap.parse_args(argv)

    from .render import quantise, write_png, write_ppm

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # before anything initialises the HIP runtime: RCCL's IPC needs dmabuf mode
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("RRTX_DIST_BACKEND", "nccl")
    device_index = local_rank % max(1, torch.cuda.device_count()) if backend == "gloo" else local_rank
    torch.cuda.set_device(device_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)
    sr = ShardedRenderer(a.scene, a.w, a.h, a.spp, a.depth, fp64=a.fp64, tile_rows=a.tile_rows, seed=a.seed, device=torch.device("cuda", device_index), use_bvh=not a.no_bvh)
    frame = sr.render()
    torch.cuda.synchronize()
    st = sr.rrt.collect()
    print("rank %d: %d rows, took %g seconds." % (sr.rank, len(sr.rows), st["kernel_ms"] / 1000.0), file=sys.stderr)
    rc = 0
    try:
        if sr.rank == 0:
            rgb = quantise(frame.cpu().numpy(), a.spp)
            if a.png:
                write_png(a.png, rgb)
            else:
                write_ppm(None, rgb)
    except Exception as e:  # the other ranks wait in the barrier below: reach it whatever happened here
        print("rank 0: could not write the image: %r" % (e,), file=sys.stderr)
        rc = 1
    finally:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    raise SystemExit(main())
