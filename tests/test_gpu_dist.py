"""Device-resident rendering through torch (the path bench.py and the multi-GPU gather use): the
framebuffer stays in HBM, the kernel runs on torch's current stream."""
import numpy as np
import pytest

from _oracle import Oracle, scene_path

pytestmark = pytest.mark.gpu


def test_render_into_a_torch_tensor_on_the_current_stream(gpu):
    import torch

    from rrt_amd.dist import ShardedRenderer

    w, h, spp = 64, 40, 4
    sr = ShardedRenderer(scene_path("final"), w, h, spp, sample_chunk=-1)
    assert sr.world == 1 and len(sr.rows) == h
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        frame = sr.render()
        side.synchronize()
    fo, _ = Oracle(scene_path("final"), w, h, False).render(spp, 50, 1984, order=1)
    assert np.array_equal(frame.cpu().numpy(), fo)
    st = sr.rrt.collect()
    assert st["renders"] == 1 and st["kernel_ms"] > 0


def test_back_to_back_launches_are_independent(gpu):
    import torch

    from rrt_amd.dist import ShardedRenderer

    sr = ShardedRenderer(scene_path("test2"), 80, 50, 5)
    first = sr.render_local().clone()
    for _ in range(3):
        sr.render_local()
    torch.cuda.synchronize()
    assert torch.equal(first, sr.local)
    st = sr.rrt.collect()
    assert st["renders"] == 4 and st["kernel_ms_sum"] >= st["kernel_ms"]
