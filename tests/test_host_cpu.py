"""Host-side logic of the product that needs no GPU: the per-epoch schedule (main_shot_vae.py:270-279,518-520), the
state_dict key layout against the REFERENCE's own key list (tests/golden/ref_state_keys.json, written by
tests/golden/make_goldens.py from the reference model for both data_parallel layouts), the checkpoint round trip in the
reference's {epoch, args, state_dict, optimizer} layout (main_shot_vae.py:237-242,202-213), and the monitor KL formula
(:330-339) against the reference's value."""
import json
import math
import os

import numpy as np
import pytest
import torch

import shot_vae_amd as S
from oracle import closed_form as C
from oracle import shotvae_oracle as O
from tests import _cases as T


def _model(name, K, dp):
    return S.VariationalAutoEncoder(name, num_input_channels=3, img_size=(32, 32), data_parallel=dp,
                                    continuous_latent_dim=128, disc_latent_dim=K, small_input=True)


def test_alpha_schedule_known_answers_and_formula():
    # alpha_schedule(0, E, a) = a e^-5, (E, E, a) = a; flat after E
    assert S.alpha_schedule(0, 200, 2.3) == pytest.approx(2.3 * math.exp(-5))
    assert S.alpha_schedule(200, 200, 2.3) == pytest.approx(2.3)
    assert S.alpha_schedule(450, 200, 2.3) == pytest.approx(2.3)
    for e in (0, 1, 10, 77, 199, 240, 399, 599):
        for E, a in ((200, 1e-3), (400, 1e-3), (150, 4.6), (240, 1.0)):
            assert S.alpha_schedule(e, E, a) == pytest.approx(a * math.exp(-5 * (1 - min(1, e / E)) ** 2), rel=1e-12)


def test_schedule_matches_reference_epoch_scalars():
    """S.schedule = the seven scalars of main_shot_vae.py:270-279 with the script's defaults (Cifar10 branch: dmi 2.3
    from :139; wmf 0.4 -> ucw over round(0.4 * epochs) epochs, :279); the Cifar100 branch overrides dmi / akb / apw
    (:161-163)."""
    for epoch in (0, 10, 150, 400, 599):
        s = S.schedule(epoch)
        want = dict(cmi=0.0, dmi=S.alpha_schedule(epoch, 200, 2.3), ew=S.alpha_schedule(epoch, 400, 1e-3),
                    kl_beta_c=S.alpha_schedule(epoch, 200, 1e-3), kl_beta_d=S.alpha_schedule(epoch, 200, 1e-3),
                    pwm=S.alpha_schedule(epoch, 200, 1.0), ucw=S.alpha_schedule(epoch, 240, 1.0))
        assert set(s) == set(want)
        for k in want:
            assert s[k] == pytest.approx(want[k], rel=1e-12), (epoch, k)
        o = O.schedule(epoch)                      # and the oracle's copy (which the GPU parity tests feed to both sides)
        for k in want:
            assert s[k] == pytest.approx(o[k], rel=1e-12), (epoch, k)
    c100 = S.schedule(10, epochs=700, dmi=4.6, akb=150, apw=400)
    assert c100["dmi"] == pytest.approx(4.6 * math.exp(-5 * (1 - 10 / 150) ** 2))
    assert c100["pwm"] == pytest.approx(math.exp(-5 * (1 - 10 / 400) ** 2))
    assert c100["ucw"] == pytest.approx(math.exp(-5 * (1 - 10 / 280) ** 2))


@pytest.mark.parametrize("name,K", [("wideresnet-28-2", 10), ("wideresnet-10-1", 10), ("wideresnet-28-10", 100)])
@pytest.mark.parametrize("dp", [False, True])
def test_state_dict_keys_equal_the_reference_key_list(name, K, dp):
    ref = json.load(open(os.path.join(T.GOLDEN, "ref_state_keys.json")))["%s|K=%d|dp=%d" % (name, K, int(dp))]
    sd = _model(name, K, dp).state_dict()
    assert [k for k, _ in ref] == list(sd.keys())
    for k, shape in ref:
        assert list(sd[k].shape) == shape, (k, list(sd[k].shape), shape)
    if dp:
        assert sum(".module." in k for k in sd) == sum(".module." in k for k, _ in ref) > 0


def test_checkpoint_round_trip_in_the_reference_layout(tmp_path):
    """A checkpoint as main_shot_vae.py:237-242 writes it (torch.optim.SGD state) resumes into FlatSGD, and FlatSGD's
    state_dict loads back into torch.optim.SGD: same parameter ids, same momentum buffers; the model accepts the other
    data_parallel key layout."""
    name, K = "wideresnet-10-1", 10
    m1 = _model(name, K, True)
    m1.load_state_dict(C.make_state(name, K=K))
    opt1 = torch.optim.SGD(m1.parameters(), lr=0.02, momentum=0.9, weight_decay=5e-4)
    m1._attach_grads()
    g = torch.Generator().manual_seed(3)
    m1._engine.grad.copy_(torch.randn(m1._engine.grad.shape, generator=g) * 1e-2)
    p0 = m1._engine.param.clone()
    opt1.step()                                        # plain torch on CPU tensors: creates the momentum buffers
    path = str(tmp_path / "checkpoint.pth.tar")
    torch.save({"epoch": 7, "args": {"lr": 0.1}, "state_dict": m1.state_dict(), "optimizer": opt1.state_dict()}, path)

    ck = torch.load(path, weights_only=False)
    m2 = _model(name, K, False)                        # the other key layout
    m2.load_state_dict(ck["state_dict"])
    opt2 = S.FlatSGD(m2)
    opt2.load_state_dict(ck["optimizer"])
    assert torch.equal(m2._engine.param, m1._engine.param) and not torch.equal(m1._engine.param, p0)
    assert opt2.param_groups[0]["lr"] == 0.02 and opt2.param_groups[0]["weight_decay"] == 5e-4
    assert opt2._steps > 0
    # momentum after the first step = g + wd * p0 in every real (un-padded) element; pads have no parameter and stay 0
    want = m1._engine.grad + 5e-4 * p0
    for prm, kind, payload in m2._views:
        got = m2._flat_view(m2._engine.mom, kind, payload)
        assert torch.allclose(got, m2._flat_view(want, kind, payload), rtol=1e-6, atol=1e-9)
    # and back: FlatSGD -> torch.optim.SGD
    sd2 = opt2.state_dict()
    opt3 = torch.optim.SGD(m1.parameters(), lr=0.5)
    opt3.load_state_dict(sd2)
    assert opt3.param_groups[0]["lr"] == 0.02 and opt3.param_groups[0]["momentum"] == 0.9
    s1, s3 = opt1.state_dict()["state"], opt3.state_dict()["state"]
    assert s1.keys() == s3.keys() and len(s1) == len(list(m1.parameters()))
    for i in s1:
        assert torch.allclose(s1[i]["momentum_buffer"], s3[i]["momentum_buffer"], rtol=1e-6, atol=1e-9)


def test_inference_kl_matches_reference_monitor():
    """S.inference_kl (main_shot_vae.py:330-339) on the oracle's unlabelled forward = the reference's Train/KL_Inference."""
    g = T.load("ref_monitor_valid_wrn10_1")
    name, K = "wideresnet-10-1", 10
    st = C.make_state(name, K=K)
    il, ll, iu, lu = C.make_batch(4, 6, K)
    nz = C.make_noise(4, 6, K)
    with torch.no_grad():
        rec, mu, ls, la = O.vae_forward(st, name, iu, nz["eps3"], u=nz["u3"])
    assert float(S.inference_kl(la, lu)) == pytest.approx(float(g["kl_inference"]), rel=2e-5)
    assert np.isfinite(float(g["kl_inference"]))


def test_grouped_step_launch_plans():
    """Which forwards of a step share a batched launch sequence (train._launch_plan): one sequence of four groups for the usual
    step; --om puts forward (4) behind the pairing kernel that needs the outputs of (3) (mixup.py:9-18); a ragged last batch
    (B_l != B_u, main_shot_vae.py:280) gives one sequence per loader.  Inside a launch the forwards whose reconstruction enters
    the loss -- (1), (3) -- come first (Engine.forward's rec_groups is a prefix), and every forward appears exactly once."""
    from shot_vae_amd.train import _launch_plan
    assert _launch_plan(False, False) == [[1, 3, 2, 4]]
    assert _launch_plan(False, True) == [[1, 3, 2], [4]]
    assert _launch_plan(True, False) == [[1, 2], [3, 4]]
    assert _launch_plan(True, True) == [[1, 2], [3], [4]]
    for ragged in (False, True):
        for om in (False, True):
            plan = _launch_plan(ragged, om)
            assert sorted(k for ids in plan for k in ids) == [1, 2, 3, 4]
            for ids in plan:
                rec = [k in (1, 3) for k in ids]
                assert rec == sorted(rec, reverse=True)                      # reconstructed forwards are a prefix
                if ragged:
                    assert len({k <= 2 for k in ids}) == 1                  # never a labelled and an unlabelled forward together
            if om:                                                           # (3) strictly before (4): the pairing needs its outputs
                i3 = [i for i, ids in enumerate(plan) if 3 in ids][0]
                i4 = [i for i, ids in enumerate(plan) if 4 in ids][0]
                assert i3 < i4


def test_flat_adam_checkpoint_reads_the_device_step_counter():
    """optim.FlatAdam.state_dict() must take `step` from the captured device counter when there is one (hipGraph replays advance
    only that): checked on the source level here (the class needs a GPU), on the device in tests/test_kernels_gpu.py."""
    import inspect
    from shot_vae_amd import optim
    src = inspect.getsource(optim.FlatAdam.state_dict)
    assert "step_dev" in src and ".item()" in src


def test_flag_fork_is_off_under_serialised_dispatch(monkeypatch):
    """The device-side fork (a kernel on the side stream waits for the data gradient's start signal) must not be used where
    kernel dispatch is serialised across streams -- rocprofv3 --pmc, AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING: the waiting
    kernel could be dispatched in front of the one it waits for.  The engine reads the environment once."""
    import shot_vae_amd.engine as E
    for var, val, want in (("ROCPROF_COUNTER_COLLECTION", "1", True), ("ROCPROF_COUNTERS", "pmc: FETCH_SIZE", True),
                           ("AMD_SERIALIZE_KERNEL", "3", True), ("HIP_LAUNCH_BLOCKING", "1", True),
                           ("AMD_SERIALIZE_KERNEL", "0", False), (None, None, False)):
        for k in ("ROCPROF_COUNTER_COLLECTION", "ROCPROF_COUNTERS", "AMD_SERIALIZE_KERNEL", "HIP_LAUNCH_BLOCKING"):
            monkeypatch.delenv(k, raising=False)
        if var:
            monkeypatch.setenv(var, val)
        monkeypatch.setattr(E, "_SERIALISED", None)
        assert E._dispatch_serialised() is want, (var, val)
    monkeypatch.setattr(E, "_SERIALISED", None)


def test_flat_sgd_is_a_torch_optimizer_and_takes_the_reference_schedule():
    """main_shot_vae.py:199,252-254: MultiStepLR over the optimizer after a linear warm-up that writes param_groups[...]['lr'] -- both
    work on FlatSGD (a torch.optim.Optimizer over model.parameters()); a scheduler's 'initial_lr' survives a checkpoint round trip."""
    m = _model("wideresnet-10-1", 10, False)
    opt = S.FlatSGD(m, lr=0.1, momentum=0.9, weight_decay=5e-4)
    assert isinstance(opt, torch.optim.Optimizer)
    assert [id(p) for p in opt.param_groups[0]["params"]] == [id(p) for p in m.parameters()]
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[2, 4], gamma=0.1)
    for g in opt.param_groups:                         # the warm-up of :223-225
        g["lr"] = 0.02
    assert opt.param_groups[0]["lr"] == 0.02
    for g in opt.param_groups:
        g["lr"] = 0.1
    lrs = []
    for _ in range(5):
        lrs.append(opt.param_groups[0]["lr"])
        sched.step()
    assert lrs == pytest.approx([0.1, 0.1, 0.01, 0.01, 0.001])
    sd = opt.state_dict()
    opt2 = S.FlatSGD(_model("wideresnet-10-1", 10, False))
    sd["param_groups"][0]["initial_lr"] = 0.1
    opt2.load_state_dict(sd)
    assert opt2.param_groups[0]["lr"] == pytest.approx(0.001) and opt2.param_groups[0]["initial_lr"] == 0.1
    assert len(opt2.param_groups[0]["params"]) == len(list(m.parameters()))
    with pytest.raises(NotImplementedError):
        opt.step(closure=lambda: None)


def test_step_logger_and_ranges(tmp_path):
    """shot_vae_amd.trace: the per-step JSONL of the loss terms (what the reference logs through TensorBoard / its progress line,
    main_shot_vae.py:367-383) and the roctx phase ranges (no-ops unless enabled; libroctx64.so loads without a GPU)."""
    path = tmp_path / "log" / "steps.jsonl"
    out = {k: torch.tensor(float(i) + 0.5) for i, k in enumerate(S.trace.SCALARS)}
    out["kl_inference"] = torch.tensor(0.25)
    out["rec1"] = torch.zeros(2, 3)                       # (tensors that are not scalars of the log are ignored)
    with S.StepLogger(str(path)) as lg:
        lg.log(out)
        rec = lg.log(out, step=7, lr=0.02)
    lines = [json.loads(ln) for ln in open(path)]
    assert len(lines) == 2 and lines[0]["step"] == 0 and lines[1]["step"] == 7 and lines[1]["lr"] == 0.02
    assert lines[1]["loss_unsup"] == 11.5 and lines[1]["kl_inference"] == 0.25 and "rec1" not in lines[1] and rec == lines[1]
    assert [k for k in lines[0] if k in S.trace.SCALARS] == list(S.trace.SCALARS)
    was = S.trace.ranges_enabled()
    try:
        S.enable_ranges(False)
        with S.step_range("forward"):
            pass
        if S.enable_ranges(True):                          # the library is in the ROCm image: a pushed range is popped again
            with S.step_range("backward"):
                with S.step_range("nested"):
                    pass
    finally:
        S.enable_ranges(was)
