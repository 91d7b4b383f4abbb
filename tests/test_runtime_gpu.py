"""Host-runtime behaviour around the kernels (round-1 advisor findings): the fused Adam's device-state mode equals the
eager mode step for step, optimizer checkpoints round-trip in and between both modes, the hipGraph sampler sees an
optimizer step in bf16 mode, the workspace never frees an address a graph may reference, a tensor on the wrong HIP
device raises, and the Trainer binds its rank's device."""
import os
import subprocess
import sys

import pytest
import torch

from _parity import record
from _util import DEV

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "image-generation-models_amd")


def _net(mode="fp32", dim=16):
    from src.models.ddpm import Unet
    torch.manual_seed(3)
    net = Unet(dim=dim, dim_mults=(1, 2), channels=3).to(DEV)
    net.compute_mode = mode
    return net


def _fake_grads(net, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    net.flat_grads.copy_(torch.randn(net.flat_grads.shape, device=DEV, generator=g) * 1e-2)


def test_adam_device_state_equals_eager_from_step_one():
    """mi_adam_step_dev evaluates 1 - b^t in double like the host does for mi_adam_step: the two modes apply the SAME update at
    t = 1, 2, 3 (with fp32 bias corrections 1 - 0.999^1 was off by 6e-5 relative)."""
    from src.runtime.optim import FlatAdam
    a, b = _net(), _net()
    oa = FlatAdam(a, lr=1e-3, betas=(0.9, 0.999))
    ob = FlatAdam(b, lr=1e-3, betas=(0.9, 0.999), device_state=True)
    for s in range(1, 4):
        _fake_grads(a, s); _fake_grads(b, s)
        oa.step(); ob.step()
        assert torch.equal(a.flat_params, b.flat_params), s
    assert ob.device_step_count() == 3
    # against torch.optim.Adam on the same gradients
    c = _net()
    ref = c.flat_params.detach().clone().requires_grad_(True)
    to = torch.optim.Adam([ref], lr=1e-3, betas=(0.9, 0.999))
    for s in range(1, 4):
        _fake_grads(c, s)
        ref.grad = c.flat_grads.clone()
        to.step()
    assert float((ref.detach() - a.flat_params).abs().max()) < 2e-7


@pytest.mark.parametrize("src_dev_state,dst_dev_state", [(False, False), (True, True), (False, True), (True, False)])
def test_flat_adam_checkpoint_round_trip(tmp_path, src_dev_state, dst_dev_state):
    """save -> load -> step continues the trajectory (moments, step count and lr restored) in and between both modes."""
    from src.runtime.optim import FlatAdam
    a = _net()
    oa = FlatAdam(a, lr=3e-4, betas=(0.9, 0.999), device_state=src_dev_state)
    for s in range(1, 4):
        _fake_grads(a, s); oa.step()
    path = str(tmp_path / "opt.pt")
    torch.save({"opt": oa.state_dict(), "w": a.flat_params.cpu()}, path)
    _fake_grads(a, 9); oa.step()                                   # the continuation to reproduce

    ck = torch.load(path)
    b = _net()
    b.flat_params.copy_(ck["w"].to(DEV)); b.mark_params_dirty()
    ob = FlatAdam(b, lr=1e-1, betas=(0.5, 0.9), device_state=dst_dev_state)      # wrong hyper-parameters on purpose
    ob.load_state_dict(ck["opt"])
    assert ob.param_groups[0]["lr"] == 3e-4 and tuple(ob.param_groups[0]["betas"]) == (0.9, 0.999)
    _fake_grads(b, 9); ob.step()
    assert ob.device_step_count() == 4 and ob.state_dict()["step"] == 4
    assert torch.equal(a.flat_params, b.flat_params)


def test_graph_sampler_sees_optimizer_step_in_bf16_mode():
    """After an optimizer step the captured denoise iteration must convolve with the NEW bf16 weight copies (they are packed
    from Python, outside the graph): graph == eager on the same noise tape, and != the pre-step images."""
    from src.models.ddpm import GaussianDiffusion
    from src.runtime.optim import FlatAdam
    from src.runtime.sampler import GraphSampler
    net = _net("bf16", dim=32)
    gd = GaussianDiffusion(net, image_size=(16, 16), timesteps=6).to(DEV)
    shape = (8, 3, 16, 16)
    g = torch.Generator().manual_seed(0)
    tape = [torch.randn(shape, generator=g) for _ in range(7)]

    def run_graph(gs):
        if gs.graph is None:
            gs._capture()
        else:
            gs.refresh()
        gs.set_image(tape[0].to(DEV)); gs.t.fill_(5)
        for i in range(6):
            gs.z.copy_(tape[1 + i].to(DEV)); gs.graph.replay()
        return gs.x.clone()

    def run_eager():
        it = iter(tape)
        gd.noise_source = lambda s, d: next(it).to(d)
        out = gd.p_sample_loop(shape, use_graph=False)
        gd.noise_source = None
        return out

    net.eval()
    gs = GraphSampler(gd, shape)
    before = run_graph(gs)
    assert float((before - run_eager()).abs().max()) < 1e-5
    net.train()
    opt = FlatAdam(net, lr=5e-2, betas=(0.9, 0.999))
    loss = gd.p_losses(torch.rand(shape, device=DEV) * 2 - 1, torch.randint(0, 6, (8,), device=DEV))
    loss.backward(); opt.step()
    net.eval()
    after_graph, after_eager = run_graph(gs), run_eager()
    assert float((after_graph - after_eager).abs().max()) < 1e-5
    assert float((after_graph - before).abs().max()) > 1e-3


def test_graph_sampler_is_recaptured_when_the_numeric_mode_changes():
    """A captured denoise iteration bakes in the kernel picks; p_sample_loop must not replay a bf16-mode graph after
    Unet.compute_mode was switched to fp32 (or the storage / fusion settings changed)."""
    from src.models.ddpm import GaussianDiffusion, Unet
    torch.manual_seed(0)
    net = Unet(dim=32, dim_mults=(1, 2), channels=3).to(DEV).eval()
    gd = GaussianDiffusion(net, image_size=(16, 16), timesteps=4).to(DEV)
    net.compute_mode = "bf16"
    gd.sample(batch_size=8)
    g1 = gd._graph
    gd.sample(batch_size=8)
    assert gd._graph is g1                                   # same settings: the graph is reused
    net.compute_mode = "fp32"
    gd.sample(batch_size=8)
    assert gd._graph is not g1
    g2 = gd._graph
    net.fuse_gn_conv = "0"
    gd.sample(batch_size=8)
    assert gd._graph is not g2


def test_workspace_growth_keeps_old_addresses_alive():
    from src.ops import functional as K
    dev = torch.device("cuda", torch.cuda.current_device())
    first = K._workspace(dev, 1024)
    ptr = first.data_ptr()
    grown = K._workspace(dev, first.numel() * 4 + (1 << 20))
    assert grown.numel() > first.numel()
    assert any(t.data_ptr() == ptr for t in K._WS_RETIRED)         # retired, not freed
    assert K._workspace(dev, 1024) is grown


def test_wrong_device_tensor_raises():
    """Every wrapper launches on the current HIP device; a tensor that lives elsewhere must not be touched from here."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from src.ops import functional as K
    x = torch.zeros(16, device="cuda:1")
    torch.cuda.set_device(0)
    with pytest.raises(RuntimeError, match="current HIP device"):
        K.mish_fwd(x)


def test_trainer_fit_binds_the_ranks_device(tmp_path):
    """`Trainer.fit` calls torch.cuda.set_device(LOCAL_RANK) before anything is launched (checked through run.py on this box's
    single GPU: LOCAL_RANK=0 and the device stays bound after fit)."""
    from src.runtime.trainer import Trainer
    from src.models.ddpm import DDPM
    dm = {"width": 8, "height": 8, "channels": 3, "transforms": {"normalize": True}}
    model = DDPM(dm, hidden_dim=8, dim_mults=(1, 2), timesteps=10, lr=1e-3, b1=0.9, b2=0.999)
    data = [(torch.rand(4, 3, 8, 8) * 2 - 1, torch.zeros(4, dtype=torch.long)) for _ in range(2)]
    called = []
    real = torch.cuda.set_device
    torch.cuda.set_device = lambda d: (called.append(torch.device(d) if not isinstance(d, int) else d), real(d))[1]
    try:
        tr = Trainer(devices=1, max_epochs=1, num_sanity_val_steps=0, enable_checkpointing=False, default_root_dir=str(tmp_path))
        tr.fit(model, train_dataloaders=data)
    finally:
        torch.cuda.set_device = real
    assert called and str(called[0]).startswith("cuda")
    assert model.denoising_model.flat_params.device.index == torch.cuda.current_device()


def test_two_gpu_nccl_fit_through_trainer(tmp_path):
    """Two ranks under torchrun, one GPU each, RCCL all-reduce from backward hooks inside Trainer.fit: both ranks end with the
    same weights.  Skipped on a one-GPU box (the driver's 8-GPU node runs it)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(PKG, "run.py"), "experiment=ddpm/synthetic", "datamodule.train_size=64",
           "datamodule.val_size=8", "datamodule.batch_size=8", "trainer.max_epochs=1", "trainer.devices=2", "model.hidden_dim=16",
           "+trainer.num_sanity_val_steps=0", "trainer.check_val_every_n_epoch=100", f"log_dir={tmp_path}", "print_config=False"]
    r = subprocess.run(cmd, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_ddpm_graphed_training_step(mode):
    """The DDPM step (randint t, randn eps, q_sample, UNet fwd, L1, UNet bwd, fused Adam with device-side step count) captured
    once and replayed as one hipGraph.  ONE replay from identical weights, Adam state and RNG state must reproduce the eager step:
    flat gradient and post-Adam weights within twice the noise two EAGER runs of that same step show (fp32 atomics of the norm
    gradients land in a different order from run to run; measured here, not assumed) -- a replay that used stale bf16 weight copies,
    a stale step count or different draws is off by orders of magnitude more.  The long replayed curve is only checked for being
    finite and falling; an eager forward after the replays must see the replayed weights (bf16 copies are repacked)."""
    from src.models.ddpm import DDPM
    from src.runtime.graphed import GraphedTrainStep
    dm = {"width": 16, "height": 16, "channels": 3, "transforms": {"normalize": True}}

    def build():
        torch.manual_seed(0)
        m = DDPM(dm, hidden_dim=32, dim_mults=(1, 2), timesteps=1000, lr=2e-3, b1=0.9, b2=0.999).to(DEV).train()
        m.denoising_model.compute_mode = mode
        m.log = lambda *a, **k: None
        o = m.configure_optimizers()
        o.device_state = True
        return m, o
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.rand(16, 3, 16, 16, device=DEV, generator=g) * 2 - 1

    def two_eager_steps():
        m, o = build()
        torch.manual_seed(11)
        for i in range(2):
            l = m.training_step((x, None), i); l.backward(); o.step()
        return m, o

    def third_step_eager():
        m, o = two_eager_steps()
        l = m.training_step((x, None), 2); l.backward(); o.step()
        torch.cuda.synchronize()
        return float(l), m.denoising_model.flat_grads.clone(), m.denoising_model.flat_params.clone()
    runs = [third_step_eager() for _ in range(5)]            # five samples of the eager step's own noise (ten pairs)
    la, ga, wa = runs[0]
    m1, o1 = two_eager_steps()
    gs = GraphedTrainStep(m1, o1, (x, None), warmup=0)       # the capture itself executes nothing
    lb = float(gs((x, None)))
    torch.cuda.synchronize()
    assert o1.device_step_count() == 3
    net = m1.denoising_model
    rel = lambda a, b: float((a - b).norm() / b.norm())      # noqa: E731
    # the noise of the eager step itself: the largest difference among five runs (a pair or two are too small a sample: the bf16
    # step amplifies the order of a few fp32 atomics chaotically, and a fourth run was seen 2.4x away from three that agreed)
    noise_g = max(rel(runs[i][1], runs[j][1]) for i in range(5) for j in range(i))
    noise_w = max(rel(runs[i][2], runs[j][2]) for i in range(5) for j in range(i))
    err_g, err_w = rel(net.flat_grads, ga), rel(net.flat_params, wa)
    floor_g, floor_w = (2e-6, 5e-7) if mode == "fp32" else (1e-4, 1e-6)      # eager runs can also agree exactly
    record(f"graph_replay_vs_eager_{mode}", grad=err_g, weights=err_w, eager_noise_grad=noise_g, eager_noise_weights=noise_w)
    assert abs(lb - la) <= 2 * max(abs(r_[0] - la) for r_ in runs[1:]) + (1e-6 if mode == "fp32" else 1e-4), (la, lb, [r_[0] for r_ in runs])
    assert err_g <= 2 * noise_g + floor_g, (err_g, noise_g)
    assert err_w <= 2 * noise_w + floor_w, (err_w, noise_w)
    # the curve: finite and falling on the fixed batch
    w_before = net.flat_params.clone()
    curve, moved = [lb], []
    for _ in range(15):
        w_prev = net.flat_params.clone()
        curve.append(float(gs((x, None))))
        moved.append(float((net.flat_params - w_prev).abs().mean()))
    assert o1.device_step_count() == 18
    # every replay applied Adam with fresh gradients: each one moves the weights by about lr per element (lr = 2e-3; a replay with a
    # stale step count, stale gradients or no optimizer launch moves them by ~0 or by a bias-correction blow-up), and fifteen steps on
    # one batch add up instead of cancelling
    assert all(2e-4 < d < 4e-3 for d in moved), moved
    assert float((net.flat_params - w_before).abs().mean()) > 3 * max(moved), (moved, float((net.flat_params - w_before).abs().mean()))
    # (every step draws fresh t / eps, so the curve is noisy: "some later step is below the first one" is the property that held on every
    #  box; "the last six are" failed once in round 4 on an unchanged fp32 path and passed on the next box)
    assert all(torch.isfinite(torch.tensor(curve))) and min(curve[1:]) < curve[0], curve
    assert sum(curve[-5:]) / 5 < 0.97 * sum(curve[:5]) / 5, curve       # ... and the averaged curve falls (lr = 2e-3 on one batch)
    # an eager forward after the replays must use the replayed weights
    net = net.eval()
    t = torch.full((16,), 10, device=DEV, dtype=torch.long)
    with torch.no_grad():
        y1 = net(x, t)
        net.mark_params_dirty()                      # forces a repack: must change nothing if the copies were current
        y2 = net(x, t)
    assert torch.equal(y1, y2)


def test_device_batch_loader_equals_host_chain():
    """HBM-resident uint8 dataset + mi_u8_gather_normalize == the per-access host chain, bit for bit: epoch coverage, ragged last
    batch, labels, ToTensor/Normalize arithmetic; flips are per-sample mirror images; data-parallel shards are disjoint."""
    import numpy as np
    from src.datamodules.base import ArrayImageDataset, DeviceBatchLoader, ShardSampler
    rng = np.random.default_rng(0)
    imgs = rng.integers(0, 256, (37, 12, 20, 3), dtype=np.uint8)
    labels = np.arange(37)
    ds = ArrayImageDataset(imgs, labels, {"convert": True, "normalize": True})
    dev = torch.device("cuda", torch.cuda.current_device())
    seen = []
    torch.manual_seed(4)
    for x, y in DeviceBatchLoader(ds, 8, dev, shuffle=True):
        assert x.is_cuda and x.dtype == torch.float32 and x.shape[1:] == (3, 12, 20)
        for xi, yi in zip(x.cpu(), y.cpu()):
            assert torch.equal(xi, ds[int(yi)][0])
            seen.append(int(yi))
    assert sorted(seen) == list(range(37)) and seen != list(range(37))
    assert [x.shape[0] for x, _ in DeviceBatchLoader(ds, 8, dev, shuffle=False)] == [8, 8, 8, 8, 5]
    # un-normalised chain
    ds01 = ArrayImageDataset(imgs, labels, {"convert": True, "normalize": False})
    x, y = next(iter(DeviceBatchLoader(ds01, 4, dev, shuffle=False)))
    assert torch.equal(x.cpu(), torch.from_numpy(imgs[:4]).permute(0, 3, 1, 2).float() / 255)
    # flip: every sample is the image or its mirror, and both occur
    dsf = ArrayImageDataset(imgs, labels, {"convert": True, "normalize": True, "flip": True})
    kinds = []
    for x, y in DeviceBatchLoader(dsf, 37, dev, shuffle=False):
        for xi, yi in zip(x.cpu(), y.cpu()):
            ref = ds[int(yi)][0]
            kinds.append(0 if torch.equal(xi, ref) else 1 if torch.equal(xi, ref.flip(-1)) else 2)
    assert set(kinds) == {0, 1}
    # shards
    a = [int(v) for _, y in DeviceBatchLoader(ds, 8, dev, True, ShardSampler(37, 0, 2)) for v in y.cpu()]
    b = [int(v) for _, y in DeviceBatchLoader(ds, 8, dev, True, ShardSampler(37, 1, 2)) for v in y.cpu()]
    assert len(a) == len(b) == 19 and set(a) | set(b) == set(range(37))


def test_device_prefetcher_delivers_the_loader_batches():
    from torch.utils.data import DataLoader, TensorDataset
    from src.datamodules.base import DevicePrefetcher
    x = torch.arange(50 * 6, dtype=torch.float32).reshape(50, 6)
    y = torch.arange(50)
    loader = DataLoader(TensorDataset(x, y), batch_size=8, pin_memory=True)
    got = list(DevicePrefetcher(loader, torch.device("cuda", torch.cuda.current_device())))
    assert len(got) == 7 and all(b[0].is_cuda for b in got)
    assert torch.equal(torch.cat([b[0].cpu() for b in got]), x) and torch.equal(torch.cat([b[1].cpu() for b in got]), y)


def _graph_node_types(g):
    """{type name: count} of a CUDAGraph captured with keep_graph=True (hipGraphGetNodes / hipGraphNodeGetType)."""
    import collections
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    names = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "waitEvent", 7: "eventRecord"}
    raw = g.raw_cuda_graph()
    n = ctypes.c_size_t(0)
    assert hip.hipGraphGetNodes(ctypes.c_void_p(raw), None, ctypes.byref(n)) == 0
    arr = (ctypes.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(ctypes.c_void_p(raw), arr, ctypes.byref(n)) == 0
    c = collections.Counter()
    for i in range(n.value):
        t = ctypes.c_int(-1)
        assert hip.hipGraphNodeGetType(ctypes.c_void_p(arr[i]), ctypes.byref(t)) == 0
        c[names.get(t.value, str(t.value))] += 1
    return dict(c)


@pytest.mark.parametrize("B", [16, 128])
def test_captured_steps_hold_kernel_nodes_only(B):
    """Round 5: a memset node (hipMemsetAsync inside the split-K GEMM) took effect out of stream order in replays of the training step.
    The captured DDPM training step and the captured denoise iteration must consist of kernel nodes only -- no memset, no memcpy -- at a
    batch that takes the generic fallbacks of the time MLP (16) and at one that takes the split GEMMs (128)."""
    from src.models.ddpm import DDPM
    from src.runtime.sampler import GraphSampler
    try:
        torch.cuda.CUDAGraph(keep_graph=True)
    except TypeError:
        pytest.skip("this torch cannot keep the captured hipGraph_t")
    torch.manual_seed(0)
    m = DDPM({"width": 16, "height": 16, "channels": 3, "transforms": {"normalize": True}}, hidden_dim=32, dim_mults=(1, 2), timesteps=1000,
             lr=1e-3, b1=0.9, b2=0.999).to(DEV).train()
    m.denoising_model.compute_mode = "bf16"
    m.log = lambda *a, **k: None
    o = m.configure_optimizers()
    o.device_state = True
    x = torch.rand(B, 3, 16, 16, device=DEV) * 2 - 1
    for i in range(2):
        l = m.training_step((x, None), i); l.backward(); o.step()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(g, stream=s):
            l = m.training_step((x, None), 0); l.backward(); o.step()
    kinds = _graph_node_types(g)
    assert set(kinds) == {"kernel"} and kinds["kernel"] > 50, kinds
    m.eval()
    gs = GraphSampler(m.diffusion_model, (8, 3, 16, 16))
    gs.refresh()
    with torch.cuda.stream(s):
        gs.t.fill_(1); gs.set_image(torch.zeros_like(gs.x)); gs._iteration()
        g2 = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(g2, stream=s):
            gs._iteration()
    torch.cuda.current_stream().wait_stream(s)
    kinds2 = _graph_node_types(g2)
    assert set(kinds2) == {"kernel"} and kinds2["kernel"] > 20, kinds2


def test_graph_replay_soak_in_fresh_processes():
    """Round 5 found a replayed-graph fault (a memset node taking effect out of stream order -> 1e38 gradients -> NaN weights) that
    showed in about one FRESH process in fifteen and in no in-process repeat, while throughput numbers looked normal.  Twelve fresh
    processes (tools/graph_soak_child.py: the cfg-2 DDPM in bf16 mode, three eager steps, capture, 30 replays each, a different
    allocator history per process): every captured step is kernel nodes only, loss / weights / gradients / Adam moments stay finite,
    every replay moves the weights by about lr (1e-4) and no gradient is absurd."""
    import json
    child = os.path.join(ROOT, "tools", "graph_soak_child.py")
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + PKG)
    bad = []
    for k in range(12):
        r = subprocess.run([sys.executable, child, "30", "16", str(k)], capture_output=True, text=True, env=env, timeout=300)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert r.returncode == 0 and lines, (k, r.stdout[-500:], r.stderr[-1500:])
        d = json.loads(lines[-1])
        ok = (d["finite"] and (d["nodes"] is None or set(d["nodes"]) <= {"kernel"}) and 2e-5 < d["min_moved"] and d["max_moved"] < 1.5e-3
              and d["max_grad"] < 1e3 and d["steps_counted"] == 33 and d["last_loss"] < d["first_loss"] * 1.5)
        if not ok:
            bad.append((k, d))
    assert not bad, bad
