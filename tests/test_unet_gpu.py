"""End-to-end parity of the HIP path (Unet / GaussianDiffusion / DDPM in
image-generation-models_amd/src/models/ddpm.py) against (a) golden vectors captured from the
reference and (b) the CPU oracle on the same seeded inputs.
Stated tolerances: epsilon-prediction rel-L2 <= 1e-4 in exact-fp32 mode (north_star; measured 3e-6); parameter gradients
rel-L2 <= 1e-3 in fp32 mode (fp32 atomics reorder sums; measured 4e-6).  The bf16-MFMA mode's tolerances are each <= 2x the
worst error MEASURED on the MI355X and recorded by these tests in profiles/r06_parity.json (rule checked by tests/test_host_cpu.py) (epsilon 0.9-1.1e-2 -> 2e-2; loss
3e-5..7e-5 -> 2e-4; per-tensor gradient rel-L2 0.05-0.10 -> 0.1-0.2; whole flat gradient 1.5e-2 -> 3e-2); the reference itself
under CPU bf16 autocast sits at 1.6e-2 on the epsilon prediction."""
import os

import numpy as np
import pytest
import torch

from _parity import record
from _util import DEV, rel_err

pytestmark = pytest.mark.gpu

# ONE fixed requirement for the bf16-MFMA mode, above the "bound <= 2x measured" regression guards: the epsilon prediction may be no
# further from the reference than the reference itself is under CPU bf16 autocast (1.6e-2 relative L2)
BF16_EPS_BUDGET = 1.6e-2


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def _tiny(golden_dir, mode="fp32"):
    from src.models.ddpm import Unet
    g = _load(golden_dir, "tiny_unet.npz")
    net = Unet(dim=8, dim_mults=(1, 2), channels=3)
    net.load_state_dict({k[2:]: _t(v) for k, v in g.items() if k.startswith("w.")})
    net.compute_mode = mode
    return g, net.to(DEV)


def test_cpu_input_fails_loudly():
    from src.models.ddpm import Unet
    net = Unet(dim=8, dim_mults=(1, 2))
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 8, 8), torch.zeros(1, dtype=torch.long))


def test_tiny_forward_golden(golden_dir):
    g, net = _tiny(golden_dir)
    net.eval()
    with torch.no_grad():
        y = net(_t(g["katA.x"]).to(DEV), _t(g["katA.t"]).to(DEV))
    assert rel_err(y, _t(g["katA.y"])) < 1e-4
    assert abs(float(y.sum()) - 46.76684601) < 2e-3            # SURVEY.md KAT-A


def test_tiny_loss_and_grads_golden(golden_dir):
    from src.models.ddpm import GaussianDiffusion
    g, net = _tiny(golden_dir)
    net.train()
    gd = GaussianDiffusion(net, image_size=(8, 8), timesteps=1000).to(DEV)
    loss = gd.p_losses(_t(g["katA.x"]).to(DEV), _t(g["katA.t"]).to(DEV), _t(g["katB.noise"]).to(DEV))
    loss.backward()
    assert abs(float(loss) - float(g["katB.loss"])) < 2e-5
    bad = []
    for k, p in net.named_parameters():
        ref = _t(g["grad." + k])
        e = rel_err(p.grad, ref) if float(ref.abs().max()) > 1e-7 else float((p.grad.cpu() - ref).abs().max())
        if e > 1e-3:
            bad.append((k, e))
    assert not bad, bad
    gd2 = GaussianDiffusion(net, image_size=(8, 8), timesteps=1000, loss_type="l2").to(DEV)
    with torch.no_grad():
        l2 = gd2.p_losses(_t(g["katA.x"]).to(DEV), _t(g["katA.t"]).to(DEV), _t(g["katB.noise"]).to(DEV))
    assert abs(float(l2) - float(g["katB.loss_l2"])) < 2e-5


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_unet_without_time_embedding_golden(golden_dir, mode):
    """Unet(with_time_emb=False) -- the reference's constructor flag that DDPM never sets (ddpm.py:186-198: no time MLP, ResnetBlocks without
    their time Linear, forward with t = None): forward, L1 p_losses and every parameter gradient against vectors produced by the reference
    itself (tools/gen_golden_notime.py); the sampler's time-bias table does not exist for it and the sampler (graph path, the default) still runs."""
    from src.models.ddpm import GaussianDiffusion, Unet
    g = _load(golden_dir, "tiny_unet_notime.npz")
    net = Unet(dim=8, dim_mults=(1, 2), channels=3, with_time_emb=False)
    assert list(net.state_dict().keys()) == [str(k) for k in g["pin.state_keys"]]
    net.load_state_dict({k[2:]: _t(v) for k, v in g.items() if k.startswith("w.")})
    net.compute_mode = mode
    net = net.to(DEV)
    x, t, noise = _t(g["katA.x"]).to(DEV), _t(g["katA.t"]).to(DEV), _t(g["katB.noise"]).to(DEV)
    net.eval()
    with torch.no_grad():
        y = net(x, t)
        y2 = net(x, torch.zeros_like(t))                     # `time` is ignored, as in the reference
    e_eps = rel_err(y, _t(g["katA.y"]))
    assert torch.equal(y, y2)
    assert e_eps < (1e-4 if mode == "fp32" else BF16_EPS_BUDGET), e_eps
    assert net.time_bias_table(1000) is None
    net.train()
    gd = GaussianDiffusion(net, image_size=(8, 8), timesteps=1000).to(DEV)
    loss = gd.p_losses(x, t, noise)
    loss.backward()
    assert abs(float(loss) - float(g["katB.loss"])) < (2e-5 if mode == "fp32" else 5e-3)
    if mode == "fp32":
        bad = []
        for k, p in net.named_parameters():
            ref = _t(g["grad." + k])
            e = rel_err(p.grad, ref) if float(ref.abs().max()) > 1e-7 else float((p.grad.cpu() - ref).abs().max())
            if e > 1e-3:
                bad.append((k, e))
        assert not bad, bad
    else:
        # (an 8-channel network has one channel per GroupNorm group: its gradients are ill-conditioned under bf16 rounding -- 0.30 on the
        #  whole gradient, measured -- so bf16 mode is held to the forward / loss bars here and to finite gradients of the right size;
        #  the gradient bars of bf16 mode are carried by the dim-32 and cfg-2 / cfg-3 models)
        gq = torch.cat([p.grad.flatten().cpu() for _, p in net.named_parameters()])
        gr = torch.cat([_t(g["grad." + k]).flatten() for k, _ in net.named_parameters()])
        assert torch.isfinite(gq).all() and 0.5 < float(gq.norm() / gr.norm()) < 2.0
    net.eval()
    gd8 = GaussianDiffusion(net, image_size=(8, 8), timesteps=8).to(DEV)
    s = gd8.sample(2)
    assert s.shape == (2, 3, 8, 8) and torch.isfinite(s).all()


def test_tiny_unet_autograd_node(golden_dir):
    """Unet.forward as a plain autograd node (loss written with torch ops by the caller)."""
    g, net = _tiny(golden_dir)
    net.train()
    x = _t(g["katA.x"]).to(DEV).requires_grad_(True)
    y = net(x, _t(g["katA.t"]).to(DEV))
    w = torch.linspace(-1, 1, y.numel(), device=DEV).reshape(y.shape)
    (y * w).sum().backward()
    from oracle import ddpm_oracle as O
    p = {k[2:]: _t(v).clone().requires_grad_(True) for k, v in g.items() if k.startswith("w.")}
    xc = _t(g["katA.x"]).clone().requires_grad_(True)
    (O.unet_forward(p, xc, _t(g["katA.t"])) * w.cpu()).sum().backward()
    assert rel_err(x.grad, xc.grad) < 1e-3
    scale = max(float(v.grad.abs().max()) for v in p.values())
    for k, q in net.named_parameters():
        r = p[k].grad
        # conv biases in front of a 1-channel-per-group GroupNorm have an exactly-zero true gradient:
        # both sides hold rounding noise there, so the bound is absolute in units of the largest gradient
        assert float((q.grad.cpu() - r).abs().max()) <= 1e-3 * float(r.abs().max()) + 2e-6 * scale, k


def test_tiny_sampler_T8_golden(golden_dir):
    from src.models.ddpm import GaussianDiffusion
    g, net = _tiny(golden_dir)
    net.eval()
    gd = GaussianDiffusion(net, image_size=(8, 8), timesteps=8).to(DEV)
    tape = iter([_t(z) for z in g["katC.tape"]])
    gd.noise_source = lambda shape, device: next(tape).to(device)
    s = gd.sample(2)
    ref = _t(g["katC.sample"])
    assert float((s.cpu() - ref).abs().max()) < 2e-4
    assert abs(float(s.sum()) + 43.85355830) < 5e-3            # SURVEY.md KAT-C


def test_offpath_posterior_helpers(golden_dir):
    """predict_start_from_noise / q_posterior / p_mean_variance / q_mean_variance (reference ddpm.py:359-388; the sampler itself
    runs the fused mi_p_sample_update kernel): (a) each equals its closed form evaluated in float64 from the schedule buffers
    (which test_host_cpu pins bit-equal to the reference's); (b) p_sample == mean + [t > 0] * exp(0.5 * log_variance) * z with the
    same z (ddpm.py:390-397), i.e. the helpers and the fused kernel describe the same posterior."""
    from src.models.ddpm import GaussianDiffusion
    g, net = _tiny(golden_dir)
    net.eval()
    gd = GaussianDiffusion(net, image_size=(8, 8), timesteps=8).to(DEV)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(4, 3, 8, 8, generator=gen).to(DEV)
    x0 = (torch.rand(4, 3, 8, 8, generator=gen) * 2 - 1).to(DEV)
    nz = torch.randn(4, 3, 8, 8, generator=gen).to(DEV)
    t = torch.tensor([5, 0, 7, 1], device=DEV)
    buf = {k: getattr(gd, k).double().cpu().numpy() for k in ("sqrt_alphas_cumprod", "alphas_cumprod", "log_one_minus_alphas_cumprod",
           "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1", "posterior_mean_coef2",
           "posterior_variance", "posterior_log_variance_clipped")}
    tc = t.cpu().numpy()
    col = lambda k: buf[k][tc].reshape(-1, 1, 1, 1)                                        # noqa: E731
    xd, x0d, nzd = (v.double().cpu().numpy() for v in (x, x0, nz))
    with torch.no_grad():
        m, v, lv = gd.q_mean_variance(x0, t)
        assert np.abs(m.cpu().numpy() - col("sqrt_alphas_cumprod") * x0d).max() < 1e-6
        assert np.abs(v.cpu().numpy() - (1.0 - col("alphas_cumprod"))).max() < 1e-6
        assert np.abs(lv.cpu().numpy() - col("log_one_minus_alphas_cumprod")).max() < 1e-5
        xs = gd.predict_start_from_noise(x, t, nz)
        want_xs = col("sqrt_recip_alphas_cumprod") * xd - col("sqrt_recipm1_alphas_cumprod") * nzd     # coefficients reach 163 at t = T - 1
        assert np.abs(xs.cpu().numpy() - want_xs).max() < 2e-6 * np.abs(col("sqrt_recip_alphas_cumprod") * xd).max()
        pm, pv, plv = gd.q_posterior(x0, x, t)
        assert np.abs(pm.cpu().numpy() - (col("posterior_mean_coef1") * x0d + col("posterior_mean_coef2") * xd)).max() < 1e-5
        assert np.abs(pv.cpu().numpy() - col("posterior_variance")).max() < 1e-7
        assert np.abs(plv.cpu().numpy() - col("posterior_log_variance_clipped")).max() < 1e-5
        mean, _, logvar = gd.p_mean_variance(x, t, clip_denoised=True)
        gd.noise_source = lambda shape, device: nz.to(device)
        xp = gd.p_sample(x, t, clip_denoised=True)
    keep = (t > 0).float().view(-1, 1, 1, 1)
    want = mean + keep * torch.exp(0.5 * logvar) * nz
    assert float((xp - want).abs().max()) < 2e-5


def test_tiny_adam5_golden(golden_dir):
    """Five optimisation steps (Adam lr 1e-4, betas .9/.999) track the reference's loss curve."""
    from src.models.ddpm import GaussianDiffusion
    from src.runtime.optim import FlatAdam
    g, net = _tiny(golden_dir)
    net.train()
    gd = GaussianDiffusion(net, image_size=(8, 8), timesteps=1000).to(DEV)
    opt = FlatAdam(net, lr=1e-4, betas=(0.9, 0.999))
    x, t, noise = _t(g["katA.x"]).to(DEV), _t(g["katA.t"]).to(DEV), _t(g["katB.noise"]).to(DEV)
    losses = []
    for _ in range(5):
        opt.zero_grad()
        loss = gd.p_losses(x, t, noise)
        loss.backward(); opt.step(); losses.append(float(loss))
    assert np.allclose(losses, g["adam5.losses"], atol=3e-5), (losses, g["adam5.losses"])
    sd = net.state_dict()
    for k in ("final_conv.1.weight", "downs.0.0.block1.block.0.weight"):
        assert float((sd[k].cpu() - _t(g["adam5.final." + k])).abs().max()) < 2e-5, k


def _seeded(dim, mults, mode):
    from src.models.ddpm import Unet
    torch.manual_seed(0)
    net = Unet(dim=dim, dim_mults=mults, channels=3)
    net.compute_mode = mode
    return net.to(DEV)


@pytest.mark.parametrize("mode,tol,gtol", [("fp32", 1e-4, 2e-3), ("bf16", 1.85e-2, 1.8e-1)])      # bf16 measured: 9.4e-3 / 9.1e-2
def test_mid_unet_golden(golden_dir, mode, tol, gtol):
    from src.models.ddpm import GaussianDiffusion
    g = _load(golden_dir, "mid_unet.npz")
    net = _seeded(32, (1, 2, 4), mode)
    x, t, noise = _t(g["x"]).to(DEV), _t(g["t"]).to(DEV), _t(g["noise"]).to(DEV)
    net.eval()
    with torch.no_grad():
        y = net(x, t)
    e_eps = rel_err(y, _t(g["y"]))
    net.train()
    gd = GaussianDiffusion(net, image_size=(16, 16), timesteps=1000).to(DEV)
    loss = gd.p_losses(x, t, noise)
    loss.backward()
    e_loss = abs(float(loss) - float(g["loss"]))
    params = dict(net.named_parameters())
    gerr = {k[5:]: rel_err(params[k[5:]].grad, _t(g[k])) for k in g if k.startswith("grad.")}
    norms = np.array([float(p.grad.double().norm()) for p in net.parameters()])
    ref = g["gradnorm_all"]
    e_norm = float(np.max(np.abs(norms - ref) / np.maximum(ref, 1e-6 + 0.01 * ref.max())))
    assert mode != "bf16" or e_eps <= BF16_EPS_BUDGET, e_eps
    record(f"mid_unet_golden_{mode}", eps_rel_l2=e_eps, loss_abs=e_loss, worst_grad_rel_l2=max(gerr.values()),
           worst_grad_key=max(gerr, key=gerr.get), worst_gradnorm_rel=e_norm,
           bounds={"eps_rel_l2": tol, "worst_grad_rel_l2": gtol, "worst_gradnorm_rel": 1e-4 if mode == "fp32" else 8.5e-2})
    assert e_eps < tol
    assert e_loss < (3e-5 if mode == "fp32" else 1e-4)                     # bf16 measured 2.6e-5
    assert max(gerr.values()) < gtol, gerr
    assert e_norm < (1e-4 if mode == "fp32" else 8.5e-2)                  # bf16 measured 4.4e-2
    ok = np.abs(norms - ref) <= gtol * 2 * np.maximum(ref, 1e-6) + 1e-7
    assert ok.all(), [(k, a, b) for (k, _), a, b, o in zip(net.named_parameters(), norms, ref, ok) if not o]


@pytest.mark.parametrize("mode,tol", [("fp32", 1e-4), ("bf16", 1.75e-2)])                     # bf16 measured: 8.9e-3
def test_cfg2_eps_prediction_golden(golden_dir, mode, tol):
    """BASELINE cfg 2 (dim 128, mults 1-2-4, 32x32): epsilon prediction vs the reference's output."""
    from src.models.ddpm import GaussianDiffusion
    g = _load(golden_dir, "cfg2_unet.npz")
    net = _seeded(128, (1, 2, 4), mode)
    gd = GaussianDiffusion(net, image_size=(32, 32), timesteps=1000).to(DEV)
    x, t, noise = _t(g["x"]).to(DEV), _t(g["t"]).to(DEV), _t(g["noise"]).to(DEV)
    xn = gd.q_sample(x, t, noise)
    assert float((xn.cpu() - _t(g["x_noisy"])).abs().max()) < 1e-6
    net.eval()
    with torch.no_grad():
        eps = net(xn, t)
    e_eps = rel_err(eps, _t(g["eps_hat"]))
    net.train()
    loss = gd.p_losses(x, t, noise)
    loss.backward()
    e_loss = abs(float(loss) - float(g["loss"]))
    norms = np.array([float(p.grad.double().norm()) for p in net.parameters()])
    ref = g["gradnorm_all"]
    e_norm = float(np.max(np.abs(norms - ref) / np.maximum(ref, 1e-6 + 0.01 * ref.max())))
    e_g1 = rel_err(dict(net.named_parameters())["final_conv.1.weight"].grad, _t(g["grad.final_conv.1.weight"]))
    e_g2 = rel_err(dict(net.named_parameters())["time_mlp.3.bias"].grad, _t(g["grad.time_mlp.3.bias"]))
    assert mode != "bf16" or e_eps <= BF16_EPS_BUDGET, e_eps
    record(f"cfg2_eps_prediction_golden_{mode}", eps_rel_l2=e_eps, loss_abs=e_loss, worst_gradnorm_rel=e_norm,
           grad_final_conv_rel_l2=e_g1, grad_time_mlp3_bias_rel_l2=e_g2, bounds={"eps_rel_l2": tol})
    assert e_eps < tol
    assert e_loss < (3e-5 if mode == "fp32" else 1e-4)                     # bf16 measured 1.6e-5
    if mode == "fp32":
        assert np.all(np.abs(norms - ref) <= 4e-3 * np.maximum(ref, 1e-6) + 1e-7)
        assert e_g1 < 2e-3
    else:
        assert e_g1 < 4e-2 and e_g2 < 6.4e-2 and e_norm < 4.4e-2          # measured 2.0e-2 / 3.2e-2 / 2.2e-2


def test_cfg2_vs_oracle_random_batch():
    """HIP vs oracle on a fresh seeded batch incl. t=0 and t=T-1 (edge timesteps)."""
    from oracle import ddpm_oracle as O
    net = _seeded(128, (1, 2, 4), "fp32").eval()
    g = torch.Generator().manual_seed(77)
    x = torch.randn(3, 3, 32, 32, generator=g)
    t = torch.tensor([0, 999, 123])
    p = {k: v.detach().cpu().contiguous() for k, v in net.state_dict().items()}
    with torch.no_grad():
        ref = O.unet_forward(p, x, t)
        y = net(x.to(DEV), t.to(DEV))
    assert rel_err(y, ref) < 1e-4


def test_full_batch_properties():
    """BASELINE size (B=128, 32x32, dim 128): size-independent properties instead of a CPU oracle run --
    batch independence (every op is per-sample: a sample's output does not depend on its batch
    mates), determinism of the forward pass, and bf16-vs-fp32 consistency."""
    net = _seeded(128, (1, 2, 4), "fp32").eval()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(128, 3, 32, 32, generator=g).to(DEV)
    t = torch.randint(0, 1000, (128,), generator=g).to(DEV)
    with torch.no_grad():
        y = net(x, t)
        y2 = net(x, t)
        ys = net(x[5:9].contiguous(), t[5:9].contiguous())
        assert torch.equal(y, y2)
        assert rel_err(ys, y[5:9]) < 1e-5
        net.compute_mode = "bf16"
        yb = net(x, t)
    e = rel_err(yb, y)
    record("cfg2_B128_forward_bf16_vs_fp32", eps_rel_l2=e, bounds={"eps_rel_l2": 2e-2})
    assert torch.isfinite(y).all() and e < 2e-2


def test_full_batch_backward_properties():
    """BASELINE size (B=128, cfg 2) through backward, in the benchmarked bf16 mode and in fp32 mode:
    (1) the gradients of a 4-sample batch equal the share those samples contribute to the 128-batch (the loss is a mean over
        B*C*H*W elements and every op is per-sample, so grad_128 = (1/32) * sum over the 32 four-sample slices -- checked for
        the sum of ALL slices, which is the whole statement);
    (2) two runs on the same inputs give gradients equal up to the order of the fp32 atomics (bias / norm-affine sums);
    (3) the bf16-mode gradient stays within the recorded distance of the fp32-mode gradient."""
    from src.models.ddpm import GaussianDiffusion
    out = {}
    g = torch.Generator().manual_seed(6)
    x = (torch.rand(128, 3, 32, 32, generator=g) * 2 - 1).to(DEV)
    t = torch.randint(0, 1000, (128,), generator=g).to(DEV)
    noise = torch.randn(128, 3, 32, 32, generator=g).to(DEV)
    grads = {}
    for mode in ("fp32", "bf16"):
        net = _seeded(128, (1, 2, 4), mode).train()
        gd = GaussianDiffusion(net, image_size=(32, 32), timesteps=1000).to(DEV)
        loss = gd.p_losses(x, t, noise); loss.backward()
        full = net.flat_grads.clone()
        loss = gd.p_losses(x, t, noise); loss.backward()
        rerun = rel_err(net.flat_grads, full)
        acc = torch.zeros_like(full)
        for i in range(0, 128, 4):
            l = gd.p_losses(x[i:i + 4].contiguous(), t[i:i + 4].contiguous(), noise[i:i + 4].contiguous()); l.backward()
            acc += net.flat_grads
        share = rel_err(acc / 32, full)
        grads[mode] = full
        out[f"{mode}_rerun_rel_l2"] = rerun
        out[f"{mode}_slices_vs_full_rel_l2"] = share
        assert torch.isfinite(full).all()
    out["bf16_vs_fp32_flat_grad_rel_l2"] = rel_err(grads["bf16"], grads["fp32"])
    record("cfg2_B128_backward_properties", bounds={"bf16_slices_vs_full_rel_l2": 1.1e-2, "bf16_vs_fp32_flat_grad_rel_l2": 1.6e-2}, **out)
    assert out["fp32_rerun_rel_l2"] < 1e-6 and out["bf16_rerun_rel_l2"] < 1e-6        # measured 1.3e-7 / 5e-8 (atomics order)
    assert out["fp32_slices_vs_full_rel_l2"] < 1e-6                                    # measured 2.6e-7
    assert out["bf16_slices_vs_full_rel_l2"] < 1.1e-2     # measured 5.5e-3: slices round their bf16 tensors independently of the full batch
    assert out["bf16_vs_fp32_flat_grad_rel_l2"] < 1.6e-2                               # measured 8.2e-3


def test_cfg3_per_gpu_batch_properties():
    """BASELINE cfg 3 at its per-GPU batch (CelebA 64x64, dim 64, mults 1-2-4-8, B = 256 / 8 = 32) in the benchmarked bf16 mode and
    in fp32 mode -- the size the data-parallel run executes per rank, where the small-grid kernel plans (64-pixel conv tiles, pixel-sliced
    LinearAttention, the uncached GroupNorm backward of the 4096-pixel slices) are the ones that run: forward determinism and batch
    independence, gradient determinism up to the order of the fp32 atomics, gradients of the four 8-sample slices summing to the
    32-batch gradient, bf16 mode within the recorded distance of fp32 mode."""
    from src.models.ddpm import GaussianDiffusion
    out = {}
    g = torch.Generator().manual_seed(16)
    x = (torch.rand(32, 3, 64, 64, generator=g) * 2 - 1).to(DEV)
    t = torch.randint(0, 1000, (32,), generator=g).to(DEV)
    noise = torch.randn(32, 3, 64, 64, generator=g).to(DEV)
    grads, eps = {}, {}
    for mode in ("fp32", "bf16"):
        net = _seeded(64, (1, 2, 4, 8), mode)
        net.eval()
        with torch.no_grad():
            y = net(x, t)
            out[f"{mode}_forward_rerun_rel_l2"] = rel_err(net(x, t), y)      # (pixel-sliced LinearAttention combines its slices with atomics)
            ys = net(x[8:16].contiguous(), t[8:16].contiguous())
        out[f"{mode}_forward_slice_vs_full_rel_l2"] = rel_err(ys, y[8:16])
        eps[mode] = y
        net.train()
        gd = GaussianDiffusion(net, image_size=(64, 64), timesteps=1000).to(DEV)
        loss = gd.p_losses(x, t, noise); loss.backward()
        full = net.flat_grads.clone()
        loss = gd.p_losses(x, t, noise); loss.backward()
        out[f"{mode}_rerun_rel_l2"] = rel_err(net.flat_grads, full)
        acc = torch.zeros_like(full)
        for i in range(0, 32, 8):
            l = gd.p_losses(x[i:i + 8].contiguous(), t[i:i + 8].contiguous(), noise[i:i + 8].contiguous()); l.backward()
            acc += net.flat_grads
        out[f"{mode}_slices_vs_full_rel_l2"] = rel_err(acc / 4, full)
        grads[mode] = full
        assert torch.isfinite(full).all() and torch.isfinite(y).all()
    out["bf16_vs_fp32_eps_rel_l2"] = rel_err(eps["bf16"], eps["fp32"])
    out["bf16_vs_fp32_flat_grad_rel_l2"] = rel_err(grads["bf16"], grads["fp32"])
    record("cfg3_B32_properties", **out)
    assert out["fp32_forward_rerun_rel_l2"] == 0.0 and out["bf16_forward_rerun_rel_l2"] == 0.0
    assert out["fp32_forward_slice_vs_full_rel_l2"] < 1e-5 and out["fp32_rerun_rel_l2"] < 1e-6 and out["bf16_rerun_rel_l2"] < 1e-6
    assert out["fp32_slices_vs_full_rel_l2"] < 1e-6
    # bf16 bounds <= 2x the values measured on the MI355X (profiles/r06_parity.json): 1.02e-2, 5.9e-3, 1.17e-2, 8.1e-3
    assert out["bf16_forward_slice_vs_full_rel_l2"] < 2e-2       # slices pick other kernel plans (tile sizes) than the full batch
    assert out["bf16_slices_vs_full_rel_l2"] < 1.2e-2
    assert out["bf16_vs_fp32_eps_rel_l2"] < 2.3e-2 and out["bf16_vs_fp32_flat_grad_rel_l2"] < 1.6e-2


def _host_tape(shape, seed, n, sha):
    """The reference's host noise tape (torch CPU generator: x_T, then one draw per reverse step); refuses any other tape."""
    import hashlib
    torch.manual_seed(seed)
    tape = [torch.randn(shape) for _ in range(n)]
    h = hashlib.sha256()
    for z in tape:
        h.update(z.numpy().tobytes())
    assert h.hexdigest() == sha, "torch's CPU generator drew a different tape than the one the golden images were made with"
    return tape


@pytest.mark.parametrize("case,hw", [("tiny", 8), ("mid", 16)])
def test_sampler_T1000_golden(golden_dir, case, hw):
    """north_star's "sampled images" parity at full length: T=1000 reverse steps (ddpm.py:399-415) on the reference's host
    noise tape, final images and the way-points after 250/500/750 steps against the reference's own sample(2) -- through the
    eager loop AND the hipGraph sampler (fp32 mode, <= 2e-4); the bf16 mode is run on the same tape and its distance recorded."""
    from src.models.ddpm import GaussianDiffusion
    from src.runtime.sampler import GraphSampler
    s = _load(golden_dir, "t1000_sampler.npz")
    T = int(s["T"])
    shape = (2, 3, hw, hw)
    tape = [z.to(DEV) for z in _host_tape(shape, int(s["seed"]), T + 1, str(s[f"{case}.tape_sha256"]))]
    marks = [int(m) for m in s["marks"]]
    if case == "tiny":
        _, net = _tiny(golden_dir)
    else:
        net = _seeded(32, (1, 2, 4), "fp32")
    net.eval()
    gd = GaussianDiffusion(net, image_size=(hw, hw), timesteps=T).to(DEV)
    ref = _t(s[f"{case}.sample"])

    def graph_run():
        it = iter(tape)
        gd.noise_source = lambda sh, d: next(it)
        way = []
        out = GraphSampler(gd, shape).run(record=way)
        gd.noise_source = None
        return out, way

    it = iter(tape)
    gd.noise_source = lambda sh, d: next(it)
    eager = gd.p_sample_loop(shape, use_graph=False)
    gd.noise_source = None
    graph, way = graph_run()
    errs = {"eager_final_max_abs": float((eager.cpu() - ref).abs().max()), "graph_final_max_abs": float((graph.cpu() - ref).abs().max())}
    for m in marks:
        errs[f"graph_after{m}_max_abs"] = float((way[m - 1].cpu() - _t(s[f"{case}.after{m}"])).abs().max())
    net.compute_mode = "bf16"
    g16, _ = graph_run()
    errs["bf16_final_rel_l2"] = rel_err(g16, ref)
    errs["bf16_final_max_abs"] = float((g16.cpu() - ref).abs().max())
    record(f"sampler_T1000_{case}", **errs)
    assert errs["eager_final_max_abs"] < 2e-5 and errs["graph_final_max_abs"] < 2e-5, errs       # measured 3.6e-6 / 4.7e-6
    assert all(errs[f"graph_after{m}_max_abs"] < 2e-5 for m in marks), errs
    assert torch.isfinite(g16).all() and float(g16.abs().max()) <= 1.0
    assert errs["bf16_final_rel_l2"] < 1e-2, errs                                                # measured 4.0e-3 / 4.9e-3


def test_fused_eval_path_matches_two_pass(golden_dir, monkeypatch):
    from src.ops import functional as _K
    monkeypatch.setattr(_K, "PW_MIN_TILES", 50)       # B = 16 here (the sampler runs B = 64): 128 tiles at 32x32, 64 at 16x16 / 256 channels
    _K.conv3x3_pw_gn_mish_picked.cache_clear()
    """Inference with GroupNorm-apply + Mish folded into block2's conv against the two-pass path, cfg 2 at B = 16, bf16 mode: the
    DEFAULT ("auto": mi_conv3x3_pw_gn_mish_sums wherever the private-weight-stream kernel takes the layer, asserted through the
    launch probe) and the two forced modes (round 2's halo kernel where the new one does not apply) give the same epsilon prediction
    up to bf16 rounding, all within the bf16 bar of the reference's output."""
    g = _load(golden_dir, "cfg2_unet.npz")
    net = _seeded(128, (1, 2, 4), "bf16").eval()
    x = _t(g["x"]).repeat(8, 1, 1, 1).to(DEV); t = _t(g["t"]).repeat(8).to(DEV)
    from src.models.ddpm import GaussianDiffusion
    gd = GaussianDiffusion(net, image_size=(32, 32), timesteps=1000).to(DEV)
    xn = gd.q_sample(x, t, _t(g["noise"]).repeat(8, 1, 1, 1).to(DEV))
    from src.ops import functional as K
    assert net.fuse_gn_conv == "auto"
    with torch.no_grad():
        K.PROBE = []
        y4 = net(xn, t)                           # the default: fused where the private-weight-stream kernel takes block2's conv
        fused_default = [q[0] for q in K.PROBE if q[0].startswith("conv_pw_kernel") and (", 3, 0, 128>" in q[0] or ", 3, 0, 64>" in q[0])]
        ln_fused = [q[0] for q in K.PROBE if q[0].startswith("ln_conv1x1_pw_kernel")]
        fin_fused = [q[0] for q in K.PROBE if q[0].startswith("small_cout_fwd_gn_kernel")]
        K.PROBE = None
        net.fuse_ln_qkv = False                   # PreNorm's LayerNorm as its own launch (what training runs) instead of inside to_qkv's staging
        y5 = net(xn, t)
        net.fuse_ln_qkv = True
        net.fuse_final = False                    # final_conv.0's GroupNorm + Mish as their own launch instead of inside final_conv.1's load
        y6 = net(xn, t)
        net.fuse_final = True
        net.fuse_gn_conv = 1
        y1 = net(xn, t)
        net.fuse_gn_conv = 2                      # ... with the GroupNorm statistics from conv1's epilogue instead of a pass over c1
        y3 = net(xn, t)
        net.fuse_gn_conv = 0
        y2 = net(xn, t)
    # every ResnetBlock whose block2 conv offers the private-weight-stream kernel PW_MIN_TILES tiles (64-pixel tiles on the small grids, round 4)
    assert len(fused_default) >= 3, fused_default
    e12, e1, e2 = rel_err(y1, y2), rel_err(y1[:2], _t(g["eps_hat"])), rel_err(y2[:2], _t(g["eps_hat"]))
    e13, e3 = rel_err(y3, y1), rel_err(y3[:2], _t(g["eps_hat"]))
    record("cfg2_fused_eval_vs_two_pass_bf16", fused_vs_two_pass_rel_l2=e12, fused_vs_reference_rel_l2=e1, two_pass_vs_reference_rel_l2=e2,
           epilogue_stats_vs_pass_stats_rel_l2=e13, epilogue_stats_vs_reference_rel_l2=e3,
           bounds={"fused_vs_two_pass_rel_l2": 1.6e-2, "fused_vs_reference_rel_l2": 2e-2, "two_pass_vs_reference_rel_l2": 2e-2,
                   "epilogue_stats_vs_reference_rel_l2": 2e-2})
    assert e12 < 1.6e-2 and e1 < 2e-2 and e2 < 2e-2
    assert e13 < 2e-2 and e3 < 2e-2                # same statistics up to fp32 summation order: bf16 rounding flips only
    e42, e4 = rel_err(y4, y2), rel_err(y4[:2], _t(g["eps_hat"]))
    record("cfg2_default_fused_eval_bf16", default_vs_two_pass_rel_l2=e42, default_vs_reference_rel_l2=e4, fused_launches=len(fused_default))
    assert e42 < 2e-2 and e4 < 2e-2
    # all six attention blocks took the LayerNorm + to_qkv kernel (mi_ln_conv1x1_pw); against the two-launch form: bf16 rounding flips only
    assert len(ln_fused) == 6, ln_fused
    e45, e5 = rel_err(y4, y5), rel_err(y5[:2], _t(g["eps_hat"]))
    record("cfg2_layernorm_in_to_qkv_bf16", fused_vs_two_launch_rel_l2=e45, two_launch_vs_reference_rel_l2=e5,
           bounds={"fused_vs_two_launch_rel_l2": 6.5e-3})                  # measured 3.3e-3 .. 4.8e-3
    assert e45 < 6.5e-3 and e5 < 2e-2
    # final_conv: the Block's GroupNorm + Mish inside the 128 -> 3 conv's load (mi_conv1x1_small_cout_gn_fwd); the two-launch form rounds the
    # normalised tensor to bf16 in between
    assert len(fin_fused) == 1, fin_fused
    e46, e6 = rel_err(y4, y6), rel_err(y6[:2], _t(g["eps_hat"]))
    record("cfg2_final_conv_fused_bf16", fused_vs_two_launch_rel_l2=e46, two_launch_vs_reference_rel_l2=e6,
           bounds={"fused_vs_two_launch_rel_l2": 3.2e-3})                  # measured 1.6e-3
    assert e46 < 3.2e-3 and e6 < 2e-2


def test_graph_sampler_matches_eager():
    """hipGraph-replayed denoise iterations == the eager loop on the same noise tape."""
    from src.models.ddpm import GaussianDiffusion
    from src.runtime.sampler import GraphSampler
    net = _seeded(32, (1, 2), "fp32").eval()
    gd = GaussianDiffusion(net, image_size=(16, 16), timesteps=6).to(DEV)
    shape = (4, 3, 16, 16)
    torch.manual_seed(1)
    tape = [torch.randn(shape, device=DEV) for _ in range(7)]
    it = iter(tape)
    gd.noise_source = lambda s, d: next(it)
    eager = gd.p_sample_loop(shape, use_graph=False)
    gs = GraphSampler(gd, shape)
    gs._capture()
    gs.set_image(tape[0]); gs.t.fill_(5)
    for i in range(6):
        gs.z.copy_(tape[1 + i]); gs.graph.replay()
    assert float((gs.x - eager).abs().max()) < 1e-5


def test_ddpm_module_steps():
    """LightningModule surface: training_step / configure_optimizers / validation_step."""
    from src.models.ddpm import DDPM
    torch.manual_seed(0)
    dm = {"width": 16, "height": 16, "channels": 3, "transforms": {"normalize": True}}
    model = DDPM(dm, hidden_dim=16, dim_mults=(1, 2), timesteps=4, lr=1e-3, b1=0.9, b2=0.999).to(DEV)
    opt = model.configure_optimizers()
    imgs = torch.rand(8, 3, 16, 16, device=DEV) * 2 - 1
    model.train()
    first = None
    for i in range(8):
        opt.zero_grad()
        loss = model.training_step((imgs, None), i)
        loss.backward(); opt.step()
        first = float(loss) if first is None else first
    assert float(loss) < first
    model.eval()
    with torch.no_grad():
        res = model.validation_step((imgs, None), 0)
    assert res.fake_image.shape == (64, 3, 16, 16) and float(res.fake_image.abs().max()) <= 1.0
    assert res.others["diffusion"].shape == imgs.shape
    sd = model.state_dict()
    assert "denoising_model.final_conv.1.weight" in sd and "diffusion_model.denoise_fn.final_conv.1.weight" in sd
    assert "diffusion_model.posterior_mean_coef2" in sd


def test_run_py_end_to_end(tmp_path):
    """python run.py experiment=ddpm/synthetic ... : compose -> train -> validate (full sampler) -> checkpoint."""
    import subprocess
    import sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "image-generation-models_amd")
    cmd = [sys.executable, os.path.join(pkg, "run.py"), "experiment=ddpm/synthetic", "model.hidden_dim=16", "model.dim_mults=[1,2]",
           "model.timesteps=8", "datamodule.train_size=256", "datamodule.val_size=64", "datamodule.batch_size=32",
           "datamodule.width=16", "datamodule.height=16", "trainer.max_epochs=2", "+trainer.precision=bf16-mixed",
           f"log_dir={tmp_path}", "seed=1", "print_config=False"]
    r = subprocess.run(cmd, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    run_dir = tmp_path / "runs" / "ddpm" / "synthetic"
    assert (run_dir / "results" / "0.jpg").exists() and (run_dir / "results" / "1.jpg").exists()
    ck = torch.load(next((run_dir / "checkpoints").glob("*.ckpt")))
    keys = set(ck["state_dict"])
    assert "denoising_model.downs.0.0.block1.block.0.weight" in keys and "diffusion_model.betas" in keys
    assert ck["state_dict"]["denoising_model.downs.0.0.block1.block.0.weight"].shape == (16, 3, 3, 3)
    lines = (run_dir / "tensorboard" / "metrics.jsonl").read_text().strip().splitlines()
    assert lines and "train_loss/loss" in lines[-1]


@pytest.mark.parametrize("mode,tol,gtol", [("fp32", 1e-4, 3e-3), ("bf16", 2e-2, 8.5e-2)])      # bf16 measured: 1.16e-2 / 4.3e-2
def test_cfg3_celeba_shape_vs_oracle(mode, tol, gtol):
    """BASELINE cfg 3 (CelebA 64x64, hidden 64, mults 1-2-4-8, 4 levels): forward, loss and gradients vs the oracle."""
    from oracle import ddpm_oracle as O
    from src.models.ddpm import GaussianDiffusion
    net = _seeded(64, (1, 2, 4, 8), mode)
    g = torch.Generator().manual_seed(11)
    B = 8
    x = torch.rand(B, 3, 64, 64, generator=g) * 2 - 1
    t = torch.tensor([0, 999, 5, 250, 500, 750, 900, 42])
    noise = torch.randn(B, 3, 64, 64, generator=g)
    p = {k: v.detach().cpu().contiguous().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    tab = O.schedule_tables(1000)
    ref_loss, ref_eps = O.p_losses(p, tab, x, t, noise)
    ref_loss.backward()
    gd = GaussianDiffusion(net, image_size=(64, 64), timesteps=1000).to(DEV)
    net.eval()
    with torch.no_grad():
        eps = net(gd.q_sample(x.to(DEV), t.to(DEV), noise.to(DEV)), t.to(DEV))
    e_eps = rel_err(eps, ref_eps)
    net.train()
    loss = gd.p_losses(x.to(DEV), t.to(DEV), noise.to(DEV))
    loss.backward()
    e_loss = abs(float(loss) - float(ref_loss))
    scale = max(float(v.grad.abs().max()) for v in p.values())
    errs = {}
    for k, q in net.named_parameters():
        r = p[k].grad
        errs[k] = float((q.grad.cpu() - r).norm()) / (float(r.norm()) + 1e-3 * scale * r.numel() ** 0.5)
    flat_ref = torch.cat([p[k].grad.flatten() for k, _ in net.named_parameters()])
    flat_got = torch.cat([q.grad.detach().cpu().flatten() for _, q in net.named_parameters()])
    assert mode != "bf16" or e_eps <= BF16_EPS_BUDGET, e_eps
    record(f"cfg3_celeba_shape_vs_oracle_{mode}", eps_rel_l2=e_eps, loss_abs=e_loss, worst_grad_rel_l2=max(errs.values()),
           worst_grad_key=max(errs, key=errs.get), whole_grad_rel_l2=rel_err(flat_got, flat_ref),
           bounds={"eps_rel_l2": tol, "worst_grad_rel_l2": gtol, "whole_grad_rel_l2": 1e-4 if mode == "fp32" else 2.6e-2})
    assert e_eps < tol
    assert e_loss < (3e-5 if mode == "fp32" else 1e-4)                     # bf16 measured 3.4e-5
    assert rel_err(flat_got, flat_ref) < (1e-4 if mode == "fp32" else 2.6e-2)   # whole gradient; bf16 measured 1.3e-2
    bad = [(k, e) for k, e in errs.items() if e > gtol]
    assert not bad, bad[:8]


def test_rccl_world1_reducer():
    """The data-parallel step on the real RCCL backend with a 1-rank group: bucketed async all-reduce of the
    flat gradient slices as backward finalises them, then Adam with the folded 1/world scale."""
    import torch.distributed as dist
    from src.models.ddpm import GaussianDiffusion
    from src.runtime.ddp import FlatGradReducer, broadcast_parameters
    from src.runtime.optim import FlatAdam
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        net = _seeded(32, (1, 2, 4), "bf16").train()
        gd = GaussianDiffusion(net, image_size=(16, 16), timesteps=1000).to(DEV)
        broadcast_parameters(net.flat_params)
        red = FlatGradReducer(net.flat_grads, bucket_bytes=1 << 20)
        net.grad_ready_hook = red.range_ready
        opt = FlatAdam(net, lr=1e-3, betas=(0.9, 0.999), grad_scale=red.grad_scale)
        x = torch.rand(8, 3, 16, 16, device=DEV) * 2 - 1
        losses = []
        for _ in range(4):
            red.begin()
            loss = gd(x)
            loss.backward()
            red.finish()
            opt.step()
            losses.append(float(loss))
        launched = red.launched
        assert launched[0][1] >= net._arch.final_range[1] and launched[-1][0] == 0 and len(launched) >= 3, (launched, net._arch.final_range, net._arch.time_range)
        assert all(a[0] == b[1] or a[0] <= b[1] for a, b in zip(launched, launched[1:]))
        assert np.isfinite(losses).all()
    finally:
        net.grad_ready_hook = None
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [5, 12])
def test_ragged_batch_sizes_vs_oracle(B):
    """Batch sizes that are not multiples of 8 / of the images-per-tile count take the fallback kernels
    (the reference's last CIFAR batch has 80 images; B=5 and 12 exercise every fallback): bf16 mode vs the oracle."""
    from oracle import ddpm_oracle as O
    from src.models.ddpm import GaussianDiffusion
    net = _seeded(64, (1, 2, 4), "bf16")
    g = torch.Generator().manual_seed(B)
    x = torch.rand(B, 3, 32, 32, generator=g) * 2 - 1
    t = torch.randint(0, 1000, (B,), generator=g)
    noise = torch.randn(B, 3, 32, 32, generator=g)
    p = {k: v.detach().cpu().contiguous().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    ref_loss, ref_eps = O.p_losses(p, O.schedule_tables(1000), x, t, noise)
    ref_loss.backward()
    gd = GaussianDiffusion(net, image_size=(32, 32), timesteps=1000).to(DEV)
    net.train()
    loss = gd.p_losses(x.to(DEV), t.to(DEV), noise.to(DEV))
    loss.backward()
    worst, wkey = 0.0, ""
    for k, q in net.named_parameters():
        r = p[k].grad
        if float(r.norm()) > 1e-4:
            e = float((q.grad.cpu() - r).norm() / r.norm())
            if e > worst:
                worst, wkey = e, k
    record(f"ragged_batch_B{B}_bf16", loss_abs=abs(float(loss) - float(ref_loss)), worst_grad_rel_l2=worst, worst_grad_key=wkey,
           bounds={"worst_grad_rel_l2": 0.14 if B == 5 else 0.12})
    assert abs(float(loss) - float(ref_loss)) < 1.5e-4                    # measured 3.7e-5 / 7.3e-5
    assert worst < (0.14 if B == 5 else 0.12), worst                      # measured 8.0e-2 (B = 5) / 6.3e-2 (B = 12)


def test_bf16_block_storage_end_to_end(golden_dir):
    """Opt-in bf16 storage of the ResnetBlock-internal tensors (block_storage="bf16"): cfg-2 epsilon prediction,
    loss and gradients stay within the bf16-mode tolerances."""
    from src.models.ddpm import GaussianDiffusion
    g = _load(golden_dir, "cfg2_unet.npz")
    net = _seeded(128, (1, 2, 4), "bf16")
    net.block_storage = "bf16"
    gd = GaussianDiffusion(net, image_size=(32, 32), timesteps=1000).to(DEV)
    B = 8
    x = _t(g["x"]).repeat(4, 1, 1, 1).to(DEV); t = _t(g["t"]).repeat(4).to(DEV); noise = _t(g["noise"]).repeat(4, 1, 1, 1).to(DEV)
    net.eval()
    with torch.no_grad():
        eps = net(gd.q_sample(x, t, noise), t)
    e_eps = rel_err(eps[:2], _t(g["eps_hat"]))
    net.train()
    loss = gd.p_losses(x, t, noise)
    loss.backward()
    e_loss = abs(float(loss) - float(g["loss"]))
    g16 = net.flat_grads.clone()
    net.block_storage = "fp32"
    loss2 = gd.p_losses(x, t, noise)
    loss2.backward()
    e_sto = rel_err(g16, net.flat_grads)
    net.compute_mode = "fp32"
    loss3 = gd.p_losses(x, t, noise)
    loss3.backward()
    e_full = rel_err(g16, net.flat_grads)
    assert e_eps <= BF16_EPS_BUDGET, e_eps
    record("cfg2_bf16_block_storage", eps_rel_l2=e_eps, loss_abs=e_loss, flat_grad_rel_l2_vs_fp32_storage=e_sto,
           flat_grad_rel_l2_vs_fp32_mode=e_full,
           bounds={"eps_rel_l2": 2.1e-2, "loss_abs": 1.4e-4, "flat_grad_rel_l2_vs_fp32_storage": 6.7e-2, "flat_grad_rel_l2_vs_fp32_mode": 7.9e-2})
    assert e_eps < 2.1e-2 and e_loss < 1.4e-4 and e_sto < 6.7e-2 and e_full < 7.9e-2       # measured 1.10e-2 / 7.1e-5 / 3.4e-2 / 4.3e-2 (round 6)


class _FixedInputsStep:
    """training_step on (x, t, eps) read from static device buffers: what GraphedTrainStep captures for the trajectory test."""

    def __init__(self, gd, t, noise):
        self.gd, self.t, self.noise = gd, t, noise

    def training_step(self, batch, i):
        return self.gd.p_losses(batch[0], self.t, self.noise)


@pytest.mark.parametrize("launch", ["eager", "graph"])
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_traj20_golden(golden_dir, mode, launch):
    """SURVEY.md section 4, integration tier: 20 Adam steps (lr 1e-4, betas .9 / .999) of the dim-32 UNet from the reference's seeded
    initialisation, a different fixed (x, t, eps) per step (tests/golden/traj20.npz, captured from the reference by
    tools/gen_golden_traj.py): the loss of EVERY step, the norm of the whole parameter vector after every step and how far a few
    tensors moved, eager and as one replayed hipGraph.  fp32 mode: per-step loss <= 1e-4 (the north-star's bar; measured ~1e-6);
    bf16 mode: within <= 2x the recorded distance (profiles/r06_parity.json)."""
    from src.models.ddpm import GaussianDiffusion
    from src.runtime.graphed import GraphedTrainStep, node_types
    from src.runtime.optim import FlatAdam
    g = _load(golden_dir, "traj20.npz")
    xs, ts, ns = _t(g["x"]).to(DEV), _t(g["t"]).to(DEV), _t(g["noise"]).to(DEV)
    steps = xs.shape[0]

    def build():
        net = _seeded(32, (1, 2, 4), mode)
        net.train()
        gd = GaussianDiffusion(net, image_size=(16, 16), timesteps=1000).to(DEV)
        return net, gd, FlatAdam(net, lr=1e-4, betas=(0.9, 0.999), device_state=(launch == "graph"))
    net, gd, opt = build()
    w0 = {k: v.detach().clone() for k, v in net.named_parameters()}
    losses, wnorm = [], []
    if launch == "graph":
        # everything lazy (packed weight tables, Adam's moments and device step count, workspaces) is created by two eager steps
        # BEFORE the capture; then weights, moments and step count go back to the reference's initial state
        init = net.flat_params.clone()
        for i in range(2):
            l = gd.p_losses(xs[0], ts[0], ns[0]); l.backward(); opt.step()
        net.flat_params.copy_(init); net.mark_params_dirty()
        for buf in opt._m + opt._v:
            buf.zero_()
        opt._step = 0
        opt._state.view(torch.int32)[0] = 0
        fx = _FixedInputsStep(gd, ts[0].clone(), ns[0].clone())
        gstep = GraphedTrainStep(fx, opt, (xs[0].clone(),), warmup=0)
        nt = node_types(gstep.graph)
        assert nt is None or set(nt) <= {"kernel"}, nt
    for k in range(steps):
        if launch == "graph":
            fx.t.copy_(ts[k]); fx.noise.copy_(ns[k])
            loss = gstep((xs[k],))
        else:
            loss = gd.p_losses(xs[k], ts[k], ns[k])
            loss.backward(); opt.step()
        losses.append(float(loss))
        wnorm.append(float(net.flat_params.double().norm()))
    e_loss = float(np.max(np.abs(np.array(losses) - g["losses"])))
    # the flat buffer holds the parameters (and alignment padding that stays zero): its norm is the reference's weight norm
    e_wn = float(np.max(np.abs(np.array(wnorm) - g["weight_norm"]) / g["weight_norm"]))
    params = dict(net.named_parameters())
    e_delta = {k[6:]: rel_err(params[k[6:]].detach() - w0[k[6:]], _t(g[k])) for k in g if k.startswith("delta.")}
    moved = float(torch.sqrt(sum(((p.detach() - w0[k]).double() ** 2).sum() for k, p in net.named_parameters())))
    e_moved = abs(moved - float(g["moved_norm"])) / float(g["moved_norm"])
    # bf16 measured on the MI355X over five runs (eager / graph): loss 4.5e-4 .. 5.6e-4, displacement 0.160 (mid_attn to_qkv) -- the two rule-bound
    # metrics (<= 2x measured); the two norm metrics are differences of large sums and move 1.7x from run to run (fp32 atomics' order):
    # recorded, asserted against fixed limits
    b = {"fp32": dict(loss=1e-4, wn=1e-6, delta=2e-2, moved=1e-3), "bf16": dict(loss=8.0e-4, wn=2e-5, delta=0.319, moved=2e-3)}[mode]
    record(f"traj20_{mode}_{launch}", worst_loss_abs=e_loss, worst_weight_norm_rel=e_wn, worst_delta_rel_l2=max(e_delta.values()),
           worst_delta_key=max(e_delta, key=e_delta.get), moved_norm_rel=e_moved,
           bounds={"worst_loss_abs": b["loss"], "worst_delta_rel_l2": b["delta"]})
    assert np.isfinite(losses).all()
    assert e_loss < b["loss"], (e_loss, losses)
    assert e_wn < b["wn"], e_wn
    assert max(e_delta.values()) < b["delta"], e_delta
    assert e_moved < b["moved"], e_moved
