"""Observation encoders of the reference (jitterbug.py:468-516 builds them, :760-761 applies them, encode_obs :927-993),
restated as plain layer lists for the GPU hook `JitterbugVecEnv.set_obs_encoder` (C ABI: jb_set_obs_encoder / jb_encode).

A layer is (W [in, out], b [out], activation).  The reference's trained weights (./autoencoder_model*.ckpt, ./VAE.pt) are
not in its repository, so the caller provides weights; the architectures are:

  * autoencoder (benchmarks/autoencoder.py:71-106): code = tanh(x @ w1 + b1), w1 [D, 12]  (feature_dimension 16 or 19);
  * VAE (benchmarks/VAE.py:15-45, 134-146; built at jitterbug.py:507-516 with data_size=19, latent_size=15):
    h = relu(fc2(relu(fc1(x)))), fc1: 19 -> 4, fc2: 4 -> 1, heads out_mean / out_std: 1 -> 15, code = mean + std * eps.

`mlp_forward` is the numpy statement of the same arithmetic (what the tests compare the kernel with)."""
import numpy as np

_ACT = {"linear": lambda x: x, "tanh": np.tanh, "relu": lambda x: np.maximum(x, 0.0)}


def autoencoder_layers(w1, b1):
    """benchmarks/autoencoder.py:71-106: the encoder half of the tied-weight autoencoder."""
    return [(np.asarray(w1, dtype=np.float32), np.asarray(b1, dtype=np.float32), "tanh")]


def vae_layers(fc1_w, fc1_b, fc2_w, fc2_b, mean_w, mean_b, std_w, std_b):
    """benchmarks/VAE.py:15-45.  Arguments are torch.nn.Linear parameters (weight [out, in], bias [out]); the two heads become
    one last layer with outputs [mean | std] (use with vae=True)."""
    t = lambda w: np.asarray(w, dtype=np.float32).T
    head_w = np.concatenate([t(mean_w), t(std_w)], axis=1)
    head_b = np.concatenate([np.asarray(mean_b, dtype=np.float32), np.asarray(std_b, dtype=np.float32)])
    return [(t(fc1_w), np.asarray(fc1_b, dtype=np.float32), "relu"), (t(fc2_w), np.asarray(fc2_b, dtype=np.float32), "relu"), (head_w, head_b, "linear")]


def random_vae_layers(data_size=19, latent_size=15, seed=0):
    """Randomly initialised VAE encoder of the reference's shape (torch.nn.Linear's default uniform(-1/sqrt(in), 1/sqrt(in)))."""
    rng = np.random.default_rng(seed)
    def lin(i, o):
        k = 1.0 / np.sqrt(i)
        return rng.uniform(-k, k, size=(o, i)), rng.uniform(-k, k, size=o)
    h1, h2 = data_size // 4, data_size // 16
    return vae_layers(*lin(data_size, h1), *lin(h1, h2), *lin(h2, latent_size), *lin(h2, latent_size))


def mlp_forward(x, layers, vae=False, eps=None):
    """Numpy reference: rows x [N, in] through the layers; vae=True returns (mean, std) unless eps [N, L] is given."""
    a = np.asarray(x, dtype=np.float32)
    for W, b, act in layers:
        a = _ACT[act]((a @ np.asarray(W, dtype=np.float32) + np.asarray(b, dtype=np.float32)).astype(np.float32))
    if not vae:
        return a
    L = a.shape[1] // 2
    mean, std = a[:, :L], a[:, L:]
    return (mean, std) if eps is None else mean + std * eps
