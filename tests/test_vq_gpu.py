"""VQ-VAE codebook step: mi_vq_nearest_fwd / mi_vq_bwd against the reference's own vectors
(tests/golden/vq_kats.npz, vqvae.py:24-43) and the CPU oracle."""
import importlib
import os

import numpy as np
import pytest
import torch

from oracle import vq_oracle as V

pytestmark = pytest.mark.gpu
K = importlib.import_module("image-generation-models_amd.src.ops.functional")


def _check_indices(idx_gpu, rows, codebook):
    """Equal to the oracle's argmin except where the two best codes are within fp32 rounding of each other
    (then the kernel's pick must be one of those two-at-rounding-distance codes)."""
    idx, best, gap = V.nearest(rows, codebook)
    got = idx_gpu.cpu().long()
    bad = (got != idx).nonzero().flatten()
    if bad.numel():
        d2 = V.squared_distances(rows[bad], codebook)
        picked = d2.gather(1, got[bad, None]).squeeze(1)
        scale = (rows[bad] ** 2).sum(1) + (codebook[got[bad]] ** 2).sum(1)
        assert bool(((picked - best[bad]) <= 4e-6 * scale).all()), "index mismatch beyond fp32 rounding"
        assert bad.numel() <= max(1, rows.shape[0] // 1000)
    return got


@pytest.mark.parametrize("tag", ["small", "cfg4", "ragged"])
def test_vq_matches_reference_vectors(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "vq_kats.npz"))
    z = torch.from_numpy(g[f"{tag}.z"])
    cb = torch.from_numpy(g[f"{tag}.codebook"])
    beta = float(g[f"{tag}.beta"])
    n, d, h, w = z.shape
    rows = V.rows_of(z).contiguous()
    rd, cd = rows.cuda(), cb.cuda()
    idx, zq, ssum = K.vq_nearest(rd, cd)
    torch.cuda.synchronize()
    assert torch.equal(idx.cpu().long(), torch.from_numpy(g[f"{tag}.idx"]))          # the spread codebooks have no near ties
    quant = zq.cpu().reshape(n, h, w, d).permute(0, 3, 1, 2)
    assert torch.equal(quant, torch.from_numpy(g[f"{tag}.quant"]))                    # a gather: bit-exact
    mse = float(ssum) / rows.numel()
    assert abs(mse - float(g[f"{tag}.vq_loss"])) <= 2e-6 * abs(float(g[f"{tag}.vq_loss"]))
    assert abs(beta * mse - float(g[f"{tag}.commit_loss"])) <= 2e-6 * abs(float(g[f"{tag}.commit_loss"]))
    # backward of vq_loss + beta * commit_loss (commit_loss already carries one factor beta, vqvae.py:39)
    dz = torch.empty_like(rd)
    dcb = torch.zeros_like(cd)
    K.vq_backward(rd, cd, idx, 1.0, beta * beta, dz=dz, dcodebook=dcb)
    torch.cuda.synchronize()
    dz_ref = V.rows_of(torch.from_numpy(g[f"{tag}.dz"]))
    assert torch.allclose(dz.cpu(), dz_ref, rtol=1e-5, atol=1e-10)
    assert torch.allclose(dcb.cpu(), torch.from_numpy(g[f"{tag}.dcodebook"]), rtol=2e-5, atol=1e-9)
    # accumulate form: dz += ...
    dz2 = dz.clone()
    K.vq_backward(rd, cd, idx, 0.0, beta * beta, dz=dz2, accumulate=True)
    assert torch.allclose(dz2.cpu(), 2 * dz_ref, rtol=1e-5, atol=1e-10)


@pytest.mark.parametrize("M,D,Kc", [(1, 4, 1), (127, 4, 3), (129, 12, 129), (1000, 128, 1000), (4096, 64, 8192), (777, 36, 257)])
def test_vq_against_oracle_shapes(M, D, Kc):
    torch.manual_seed(M + D + Kc)
    rows = torch.randn(M, D)
    cb = torch.randn(Kc, D) * 0.9
    idx, zq, ssum = K.vq_nearest(rows.cuda(), cb.cuda())
    got = _check_indices(idx, rows, cb)
    assert torch.equal(zq.cpu(), cb[got])
    ref = float(((rows - cb[got]).double() ** 2).sum())
    assert abs(float(ssum) - ref) <= 1e-5 * ref + 1e-12


def test_vq_strided_rows_and_exact_ties():
    torch.manual_seed(3)
    M, D, Kc = 300, 16, 64
    buf = torch.randn(M, 40, device="cuda")
    rows = buf[:, 8:8 + D]                                   # row stride 40, offset 8 floats (16-byte aligned)
    cb = torch.randn(Kc, D)
    cb[17] = cb[5]; cb[40] = cb[5]                            # exact duplicates: the lowest index must win (torch.argmin)
    idx, zq, _ = K.vq_nearest(rows, cb.cuda())
    ref_idx, _, _ = V.nearest(rows.cpu(), cb)
    got = idx.cpu().long()
    assert not bool(((got == 17) | (got == 40)).any())
    assert torch.equal(got, ref_idx)
    assert torch.equal(zq.cpu(), cb[got])


def test_vq_full_size_properties():
    """B=256 VQ-VAE batch worth of latents (256 x 32 x 32 rows, D=64, K=512): properties that need no CPU pass over the whole matrix."""
    torch.manual_seed(11)
    M, D, Kc = 256 * 32 * 32, 64, 512
    cb = torch.randn(Kc, D, device="cuda")
    rows = torch.randn(M, D, device="cuda") * 0.7
    idx, zq, ssum = K.vq_nearest(rows, cb)
    # idempotence: codebook rows quantise to themselves with zero loss
    idx2, zq2, ssum2 = K.vq_nearest(zq, cb)
    assert torch.equal(idx2, idx) and torch.equal(zq2, zq) and float(ssum2) == 0.0
    # optimality: no other code is closer (checked with torch on the device, chunked)
    mine = ((rows - zq) ** 2).sum(1)
    for s in range(0, M, 4096):
        d2 = ((rows[s:s + 4096, None, :] - cb[None]) ** 2).sum(-1)                   # direct differences, no expansion
        assert bool((mine[s:s + 4096] <= d2.min(1).values * (1 + 1e-5) + 1e-6).all())
    assert abs(float(ssum) - float(mine.double().sum())) <= 1e-5 * float(ssum)
    # a sampled slice against the CPU oracle
    _check_indices(idx[:8192], rows[:8192].cpu(), cb.cpu())
    # codebook gradient conserves mass: sum_k dE[k] = -(sum_m dz_m) when g_vq == g_commit
    dz = torch.empty_like(rows); dcb = torch.zeros_like(cb)
    K.vq_backward(rows, cb, idx, 1.0, 1.0, dz=dz, dcodebook=dcb)
    assert torch.allclose(dcb.sum(0), -dz.sum(0), rtol=1e-3, atol=1e-7)
    drows, dcode = V.vq_backward(rows[:4096].cpu(), cb.cpu(), idx[:4096].cpu().long(), 1.0, 1.0)
    assert torch.allclose(dz[:4096].cpu() * (M / 4096), drows, rtol=1e-5, atol=1e-10)


@pytest.mark.parametrize("tag", ["small", "cfg4", "ragged"])
def test_vector_quantizer_module_matches_reference(golden_dir, tag):
    """The host-side mirror (src/models/vqvae.py) driven exactly like VQVAE.training_step drives the reference class."""
    VQ = importlib.import_module("image-generation-models_amd.src.models.vqvae")
    g = np.load(os.path.join(golden_dir, "vq_kats.npz"))
    cb = torch.from_numpy(g[f"{tag}.codebook"])
    beta = float(g[f"{tag}.beta"])
    vq = VQ.VectorQuantizer(cb.shape[0], cb.shape[1], beta).cuda()
    assert tuple(vq.state_dict().keys()) == ("embedding",)
    lim = 1.0 / cb.shape[0]
    assert float(vq.embedding.detach().abs().max()) <= lim and float(vq.embedding.detach().std()) > 0.4 * lim      # uniform(-1/K, 1/K), vqvae.py:16-19
    vq.load_state_dict({"embedding": cb})
    z = torch.from_numpy(g[f"{tag}.z"]).cuda().requires_grad_(True)
    quant, vq_loss, commit_loss = vq(z)
    assert torch.equal(quant.detach().cpu(), torch.from_numpy(g[f"{tag}.quant"]))
    assert abs(float(vq_loss) - float(g[f"{tag}.vq_loss"])) <= 2e-6 * abs(float(g[f"{tag}.vq_loss"]))
    assert abs(float(commit_loss) - float(g[f"{tag}.commit_loss"])) <= 2e-6 * abs(float(g[f"{tag}.commit_loss"]))
    (vq_loss + beta * commit_loss).backward()
    assert torch.allclose(z.grad.cpu(), torch.from_numpy(g[f"{tag}.dz"]), rtol=1e-5, atol=1e-10)
    assert torch.allclose(vq.embedding.grad.cpu(), torch.from_numpy(g[f"{tag}.dcodebook"]), rtol=2e-5, atol=1e-9)
    assert torch.equal(vq.indices(z.detach()).flatten().cpu().long(), torch.from_numpy(g[f"{tag}.idx"]))
    # quant_z differentiated directly (the class does not detach it): the gather's gradient reaches the codebook
    vq.embedding.grad = None
    q2, _, _ = vq(z.detach())
    wgt = torch.randn_like(q2)
    (q2 * wgt).sum().backward()
    idx = torch.from_numpy(g[f"{tag}.idx"])
    ref = torch.zeros_like(cb).index_add_(0, idx, V.rows_of(wgt.cpu()))
    # (the bound is relative to the sum of the MAGNITUDES of a code's rows: the kernel adds them with fp32 atomics in whatever order they
    #  arrive, and a sum that cancels to ~0 keeps the rounding of its summands)
    mag = torch.zeros_like(cb).index_add_(0, idx, V.rows_of(wgt.cpu()).abs())
    assert bool(((vq.embedding.grad.cpu() - ref).abs() <= 2e-6 * mag + 1e-6).all())


def test_vector_quantizer_refuses_cpu():
    VQ = importlib.import_module("image-generation-models_amd.src.models.vqvae")
    with pytest.raises(RuntimeError):
        VQ.VectorQuantizer(16, 8, 0.25)(torch.randn(1, 8, 2, 2))
