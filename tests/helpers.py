"""Shared test helpers: deterministic weights identical to tests/golden/gen_golden.py:fill_params."""
import numpy as np
import torch

F32 = np.float32


def vae_param_spec(x_dim=784, z_dim=40, hidden=500):
    """named_parameters() order of ELBO(Generator, Variational) in
    examples/variational_autoencoder/vae_mnist.py:16-86 (generator registered first)."""
    return [
        ("generator.sequential.0.weight", (hidden, z_dim)), ("generator.sequential.0.bias", (hidden,)),
        ("generator.sequential.2.weight", (hidden, hidden)), ("generator.sequential.2.bias", (hidden,)),
        ("generator.sequential.4.weight", (x_dim, hidden)), ("generator.sequential.4.bias", (x_dim,)),
        ("variational.sq.0.weight", (hidden, x_dim)), ("variational.sq.0.bias", (hidden,)),
        ("variational.sq.2.weight", (hidden, hidden)), ("variational.sq.2.bias", (hidden,)),
        ("variational.fc3.weight", (z_dim, hidden)), ("variational.fc3.bias", (z_dim,)),
        ("variational.fc4.weight", (z_dim, hidden)), ("variational.fc4.bias", (z_dim,)),
    ]


def iwae_param_spec(x_dim=784, z_dim=40, hidden=500):
    """named_parameters() order of ImportanceWeightedObjective(Generator, Variational),
    examples/variational_autoencoder/iwae.py:34-120."""
    return [
        ("generator.gen_sq.0.weight", (hidden, z_dim)), ("generator.gen_sq.0.bias", (hidden,)),
        ("generator.gen_sq.2.weight", (hidden, hidden)), ("generator.gen_sq.2.bias", (hidden,)),
        ("generator.gen_sq.4.weight", (x_dim, hidden)), ("generator.gen_sq.4.bias", (x_dim,)),
        ("variational.output_logits.0.weight", (hidden, x_dim)), ("variational.output_logits.0.bias", (hidden,)),
        ("variational.output_logits.2.weight", (hidden, hidden)), ("variational.output_logits.2.bias", (hidden,)),
        ("variational.output_mean.weight", (z_dim, hidden)), ("variational.output_mean.bias", (z_dim,)),
        ("variational.output_logstd.weight", (z_dim, hidden)), ("variational.output_logstd.bias", (z_dim,)),
    ]


def make_params(spec, seed, requires_grad=True, device="cpu"):
    """U(-1/sqrt(fan_in), 1/sqrt(fan_in)) from RandomState(seed), drawn in spec order."""
    rng = np.random.RandomState(seed)
    out = {}
    for name, shp in spec:
        fan_in = shp[-1] if len(shp) > 1 else shp[0]
        s = 1.0 / np.sqrt(max(fan_in, 1))
        v = torch.tensor(rng.uniform(-s, s, size=shp).astype(F32), device=device)
        v.requires_grad_(requires_grad)
        out[name] = v
    return out


def load_params_into(module, seed):
    """Same numbers, copied into an nn.Module in named_parameters() order."""
    rng = np.random.RandomState(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            shp = tuple(p.shape)
            fan_in = shp[-1] if len(shp) > 1 else shp[0]
            s = 1.0 / np.sqrt(max(fan_in, 1))
            p.copy_(torch.tensor(rng.uniform(-s, s, size=shp).astype(F32)).to(p.device))


def vae_data(B, x_dim=784, z_dim=40):
    rng = np.random.RandomState(600 + B)
    x = (rng.uniform(size=(B, x_dim)) < 0.5).astype(F32)
    e1 = rng.standard_normal((B, z_dim)).astype(F32)
    e2 = rng.standard_normal((B, z_dim)).astype(F32)
    return x, e1, e2


def iwae_data(B, K, x_dim=784, z_dim=40):
    rng = np.random.RandomState(700 + B + K)
    x = (rng.uniform(size=(B, x_dim)) < 0.5).astype(F32)
    e1 = rng.standard_normal((K, B, z_dim)).astype(F32)
    e2 = rng.standard_normal((K, B, z_dim)).astype(F32)
    return x, e1, e2


def bnn_data(B, K):
    rng = np.random.RandomState(800 + B + K)
    x = rng.standard_normal((B, 13)).astype(F32)
    y = rng.standard_normal((B,)).astype(F32)
    eps = []
    for _ in range(2):
        eps.append(rng.standard_normal((K, 50, 14)).astype(F32))
        eps.append(rng.standard_normal((K, 1, 51)).astype(F32))
    return x, y, eps


def bnn_params(B, K, device="cpu"):
    prng = np.random.RandomState(3000 + B + K)
    shapes = [(50, 14), (1, 51)]
    w_means = [torch.tensor((0.1 * prng.standard_normal(s)).astype(F32), device=device) for s in shapes]
    w_logstds = [torch.tensor((-1.0 + 0.1 * prng.standard_normal(s)).astype(F32), device=device) for s in shapes]
    y_logstd = torch.full([1], 0.3, dtype=torch.float32, device=device)
    for t in w_means + w_logstds + [y_logstd]:
        t.requires_grad_(True)
    return w_means, w_logstds, y_logstd


def grad_stats(named_grads):
    norms, sums = [], []
    for _, g in named_grads:
        norms.append(float(g.double().norm()))
        sums.append(float(g.double().sum()))
    return np.array(norms), np.array(sums)
