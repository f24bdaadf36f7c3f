"""CPU: the output-slab layout (obs | reward | terminated | truncated in one allocation) and its typed views."""
import numpy as np
import torch

from gym_continuousdoubleauction_amd.parallel import slab_layout, slab_views


def test_layout_is_aligned_and_disjoint():
    for n, od, a in ((1, 168, 4), (7, 168, 5), (4096, 168, 4), (2048, 168, 8), (3, 42, 16)):
        lay = slab_layout(n, od, a)
        assert lay["obs"] == 0 and lay["reward"] % 8 == 0 and lay["bytes"] % 16 == 0
        assert lay["reward"] >= n * od * 4 and lay["terminated"] == lay["reward"] + n * a * 8
        assert lay["truncated"] == lay["terminated"] + n and lay["bytes"] >= lay["truncated"] + n


def test_views_alias_the_slab_without_copies():
    n, od, a, world = 5, 168, 4, 3
    lay = slab_layout(n, od, a)
    g = torch.zeros((world, lay["bytes"]), dtype=torch.uint8)
    obs, rew, term, trunc = slab_views(g, lay)
    assert obs.shape == (world, n, od) and rew.shape == (world, n, a) and term.shape == (world, n) and trunc.shape == (world, n)
    assert obs.dtype == torch.float32 and rew.dtype == torch.float64
    obs[1, 2, 3] = 1.5
    rew[2, 4, 1] = -2.25
    term[0, 4] = 1
    trunc[2, 0] = 1
    raw = g.numpy()
    assert raw[1, (2 * od + 3) * 4:(2 * od + 3) * 4 + 4].view(np.float32)[0] == 1.5
    assert raw[2, lay["reward"] + (4 * a + 1) * 8:lay["reward"] + (4 * a + 1) * 8 + 8].view(np.float64)[0] == -2.25
    assert raw[0, lay["terminated"] + 4] == 1 and raw[2, lay["truncated"]] == 1
    one = slab_views(g[1], lay)
    assert one[0].shape == (n, od) and one[0][2, 3] == 1.5 and one[0].data_ptr() == g[1].data_ptr()
