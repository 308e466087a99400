"""Observation-encoder hook (reference jitterbug.py:760-761, 927-993, benchmarks/autoencoder.py:71-106,
benchmarks/VAE.py:15-45, 134-146).  The reference's trained weights are not in its repository, so there are no golden
codes: the kernel is compared with a plain fp32 restatement of the same dense layers (numpy, and torch.nn.Linear for the
VAE trunk), tolerance 1e-5 .. 3e-5 absolute on tanh/relu outputs of O(1) magnitude."""
import numpy as np
import pytest

from jitterbug_amd import encoders, model


def test_numpy_reference_matches_torch_modules():
    """encoders.vae_layers / mlp_forward restate torch's Linear + relu stack of benchmarks/VAE.py:15-45."""
    import torch
    torch.manual_seed(0)
    fc1, fc2 = torch.nn.Linear(19, 4), torch.nn.Linear(4, 1)
    om, os_ = torch.nn.Linear(1, 15), torch.nn.Linear(1, 15)
    x = torch.randn(64, 19)
    with torch.no_grad():
        h = torch.relu(fc2(torch.relu(fc1(x))))
        mean_t, std_t = om(h).numpy(), os_(h).numpy()
    g = lambda m: (m.weight.detach().numpy(), m.bias.detach().numpy())
    layers = encoders.vae_layers(*g(fc1), *g(fc2), *g(om), *g(os_))
    mean, std = encoders.mlp_forward(x.numpy(), layers, vae=True)
    assert np.abs(mean - mean_t).max() < 1e-6 and np.abs(std - std_t).max() < 1e-6
    assert [w.shape for w, _, _ in layers] == [(19, 4), (4, 1), (1, 30)]


@pytest.mark.gpu
@pytest.mark.parametrize("task,width", [("face_direction", 16), ("move_to_pose", 19)])
def test_autoencoder_hook_matches_reference_arithmetic(task, width):
    from jitterbug_amd.vec_env import JitterbugVecEnv
    n = 777
    rng = np.random.default_rng(0)
    layers = encoders.autoencoder_layers(rng.normal(0, 1, size=(width, 12)), rng.normal(0, 0.1, size=12))
    env = JitterbugVecEnv(n, task, seed=2)
    assert env.encoded_dim == 0
    with pytest.raises(RuntimeError):
        env.encode(np.zeros((n, width), dtype=np.float32))          # fails loudly without an encoder
    env.set_obs_encoder(layers)
    assert env.encoded_dim == 12
    obs = env.reset()
    for _ in range(3):
        obs, _, _, _ = env.step(rng.uniform(-1, 1, size=n).astype(np.float32))
    code = env.encode(obs)
    W, b, _ = layers[0]
    ref = np.tanh(obs.astype(np.float64) @ W.astype(np.float64) + b.astype(np.float64))       # exact arithmetic of the same layer
    err = np.abs(code - ref).max()
    print("autoencoder hook max abs err %.2e" % err)
    assert code.shape == (n, 12) and err < 3e-5           # fp32 sums of 16-19 terms with O(1) weights
    env.set_obs_encoder(None)
    assert env.encoded_dim == 0
    with pytest.raises(RuntimeError):
        env.set_obs_encoder(encoders.autoencoder_layers(np.zeros((width + 1, 12)), np.zeros(12)))     # wrong input width
    env.close()


@pytest.mark.gpu
def test_vae_hook_mean_std_and_sampling():
    from jitterbug_amd.vec_env import JitterbugVecEnv
    n, task = 4096, "move_to_pose"
    layers = encoders.random_vae_layers(19, 15, seed=3)
    # make the heads input-dependent enough to be a real test: widen the bottleneck's scale
    env = JitterbugVecEnv(n, task, seed=5)
    obs = env.reset()
    mean, std = encoders.mlp_forward(obs, layers, vae=True)
    env.set_obs_encoder(layers, vae=True)
    assert env.encoded_dim == 15
    z1, z2 = env.encode(obs), env.encode(obs)
    assert not np.array_equal(z1, z2)                              # a fresh noise draw per call, like the reference
    eps = (z1 - mean) / np.where(np.abs(std) > 1e-6, std, 1.0)
    ok = np.abs(std) > 1e-3
    e = eps[ok]
    assert abs(e.mean()) < 0.02 and abs(e.std() - 1.0) < 0.02 and np.abs(e).max() < 6.5     # eps ~ N(0, 1)
    assert abs(np.mean(e ** 3)) < 0.1 and abs(np.mean(e ** 4) - 3.0) < 0.2
    # deterministic part: with the std head zeroed the code is the mean
    W, b, act = layers[-1]
    W0, b0 = W.copy(), b.copy(); W0[:, 15:] = 0; b0[15:] = 0
    env.set_obs_encoder(layers[:-1] + [(W0, b0, act)], vae=True)
    assert np.abs(env.encode(obs) - mean).max() < 1e-5
    # same seed, same env indices, same call number -> same noise (reproducible streams)
    env2 = JitterbugVecEnv(n, task, seed=5); env2.reset(); env2.set_obs_encoder(layers, vae=True)
    env.set_obs_encoder(layers, vae=True)
    assert np.array_equal(env.encode(obs), env2.encode(obs))
    env.close(); env2.close()


@pytest.mark.gpu
def test_environment_with_encoder_returns_observations_key():
    from jitterbug_amd import suite
    layers = encoders.autoencoder_layers(np.random.default_rng(1).normal(size=(16, 12)), np.zeros(12))
    env = suite.load("jitterbug", "face_direction", environment_kwargs=dict(obs_encoder=layers))
    ts = env.reset()
    assert list(ts.observation.keys()) == ["observations"] and ts.observation["observations"].shape == (12,)
    assert env.observation_spec()["observations"].shape == (12,)
    ts = env.step([0.5])
    assert ts.observation["observations"].shape == (12,) and np.all(np.abs(ts.observation["observations"]) <= 1.0)
    env.close()
