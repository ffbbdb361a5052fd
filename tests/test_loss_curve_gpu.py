"""Loss-curve overlay gate (BASELINE.json north_star): 30 SGD steps of the SHOT-VAE loop on WRN-28-2 / K = 10
at B_l = B_u = 64 from one initialisation with identical scripted noise -- the fp32 CPU oracle, the HIP path in fp32-operand
mode and in bf16 (sequential and grouped schedule).  tools/loss_curve.py writes the same curves (60 steps) to
profiles/r02_loss_curves.jsonl.

What can be demanded: training is a chaotic map of its rounding errors -- the fp32 HIP path itself drifts from the fp32
oracle by 1-2 % in the total losses and by up to ~20 % in the small terms (KL_d of the unlabelled batch, the posterior
terms) within 60 steps -- so the gate is (a) absolute bounds on the terms that dominate the objective and (b) bf16 no
further from the oracle than a small multiple of what fp32 rounding alone produces."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.timeout(900)
def test_bf16_loss_curve_overlays_fp32_oracle():
    import loss_curve as LC
    from oracle import shotvae_oracle as O
    name, K, B, steps, lr = "wideresnet-28-2", 10, 64, 30, 0.02
    sch = O.schedule(10)
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    ref = LC.run_oracle(name, K, B, steps, lr, sch)
    dev = {"fp32": LC.deviations(LC.run_hip(name, K, B, steps, lr, sch, "fp32"), ref),
           "bf16": LC.deviations(LC.run_hip(name, K, B, steps, lr, sch, "bf16"), ref),
           "bf16_grouped": LC.deviations(LC.run_hip(name, K, B, steps, lr, sch, "bf16", grouped=True), ref)}
    print("\n" + "\n".join("%-13s %s" % (r, {k: round(v, 4) for k, v in d.items()}) for r, d in dev.items()))
    assert ref[-1]["loss_sup"] < 0.5 * ref[0]["loss_sup"], "the run must actually train"
    for run in ("bf16", "bf16_grouped"):
        d = dev[run]
        assert d["recon_l"] < 2e-3 and d["recon_u"] < 2e-3, (run, d)           # the reconstruction terms (dominant)
        assert d["loss_sup"] < 5e-2 and d["loss_unsup"] < 6e-2, (run, d)       # the two objectives
        for k, v in d.items():
            # (the small terms -- KL_d, the posterior terms -- drift 0.05-0.39 within 30 steps across repeats and kernel sets that
            #  differ in rounding only, fp32-operand runs included: tools/probes/loss_curve_kernel_ab.py; their cap says "same
            #  order of magnitude", the terms that dominate the objective are held to 2e-3 / 5e-2 above)
            assert v < (0.5 if (k.startswith("kld") or "_post_" in k) else 0.35), (run, k, v)
            # no worse than fp32 rounding drift x3.  The fp32 run's own drift is ONE draw of a chaotic map (float-atomic order):
            # on the small terms (KL_d, the posterior terms: up to ~20 % within 60 steps, see above) a lucky draw of 4 % made
            # this ratio gate fail a bf16 run at 16 % -- their reference drift is floored.  Round 5 measured the spread directly
            # (tools/probes/loss_curve_kernel_ab.py: the same 30 steps under four kernel sets and as repeats): KL_d of the unlabelled
            # batch 0.05-0.10 for the fp32-operand run and 0.05-0.28 for bf16, the continuous posterior term 0.08-0.28 for fp32 --
            # repeats of ONE configuration differ as much as different kernel sets do (the weight gradients' float atomics: 1e-7
            # per step, amplified).  Floor = 0.10: the ratio gate then sits just inside the absolute one (0.35) for those terms.
            small = k.startswith("kld") or "_post_" in k
            assert v <= 3.0 * max(dev["fp32"][k], 0.10 if small else 0.0) + 2e-2, (run, k, v, dev["fp32"][k])
    # Gates that do NOT depend on chaotic drift: the FIRST step of every run -- no trajectory yet -- at the single-step
    # parity gates of SURVEY.md 8d: 1e-3 in fp32-operand mode, 5e-3 in bf16, sequential and grouped schedule.  The posterior terms are
    # squared DIFFERENCES between the outputs of two forwards (main_shot_vae.py:319-321): at this harness's default initialisation the
    # labelled continuous one reads 1.19e-2 off the oracle in bf16 (measured, round 6; 2-5e-3 on the closed-form weights of
    # test_step_matches_oracle_b64) -- gated at 2e-2 here
    first = LC.run_hip(name, K, B, 1, lr, sch, "fp32")[0]
    for k in LC.TERMS:
        assert abs(first[k] - ref[0][k]) <= 1e-3 * max(abs(ref[0][k]), 1e-6), (k, first[k], ref[0][k])
    for grouped in (False, True):
        first = LC.run_hip(name, K, B, 1, lr, sch, "bf16", grouped=grouped)[0]
        for k in LC.TERMS:
            tk = 2e-2 if "_post_" in k else 5e-3
            assert abs(first[k] - ref[0][k]) <= tk * max(abs(ref[0][k]), 1e-6), ("bf16 first step", grouped, k, first[k], ref[0][k])


@pytest.mark.timeout(900)
def test_bf16_gradient_at_a_conditioned_point():
    """a gradient-fidelity check at a CONDITIONED point.  The bf16 gradient gates of the step-level tests are wide
    (cosine > 0.914, relative L2 < 0.416) because their fixtures sit at the ill-conditioned initial weights.  Here: default
    initialisation, 60 SGD steps of the fp32 oracle (the loss-curve run: loss_sup 2.34 -> 0.46), then ONE step from that state -- the
    HIP path (sequential and grouped: the timed path's kernels incl. the fused backward) against the oracle's backward on the same
    batch and noise.
      * fp32-operand mode: flat-gradient cosine >= 0.9999, relative L2 <= 1.5e-2 -- the kernels compute the reference's gradient;
      * bf16: what bf16 OPERANDS cost on this network is measured on the reference arithmetic itself -- torch's CPU bf16 autocast of
        the oracle against its own fp32 gradient at this very point: cosine 0.9527, relative L2 0.309
        (tools/probes/autocast_conditioned.py; 0.899 / 0.450 at the initial weights).  The verdict's hoped-for 0.99 / 0.1 is not what
        bf16 gives here; the HIP path must beat the autocast numbers: cosine >= 0.955, relative L2 <= 0.30 (measured 0.964 / 0.269),
        per-tensor norm ratios of every conv / linear weight within 10 %."""
    import loss_curve as LC
    import shot_vae_amd as S
    from oracle import closed_form as C
    from oracle import shotvae_oracle as O
    from tests import _cases as T
    name, K, B, steps, lr = "wideresnet-28-2", 10, 64, 60, 0.02
    sch = O.schedule(10)
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    curve, st = LC.run_oracle(name, K, B, steps, lr, sch, return_state=True)
    assert curve[-1]["loss_sup"] < 0.3 * curve[0]["loss_sup"]
    il, ll, iu, lu = LC.batches(4, B, K)[steps % 4]
    nz = C.make_noise(B, B, K, stream0=9000 + 100 * steps)
    ost = {k: v.clone() for k, v in st.items()}
    for k in ost:
        if O.is_param(k):
            ost[k].requires_grad_(True)
    orc = O.train_step(ost, name, il, ll, iu, nz, sch)
    for dtype, grouped in (("fp32", False), ("bf16", False), ("bf16", True)):
        model = S.VariationalAutoEncoder(name, num_input_channels=3, img_size=(32, 32), data_parallel=True, continuous_latent_dim=128,
                                         disc_latent_dim=K, small_input=True, compute_dtype=dtype)
        model.load_state_dict({k: v.detach() for k, v in st.items()})
        model = model.cuda().train()
        elbo, cls = S.VAECriterion(discrete_dim=K, bce_reconstruction=True).cuda(), S.ClsCriterion()
        S.FlatSGD(model).zero_grad()
        step = S.train_step_grouped if grouped else S.train_step
        with T.rng_for_step(nz):
            out = step(model, elbo, cls, None, il.cuda(), ll.cuda(), iu.cuda(), sch, return_outputs=True)
        torch.cuda.synchronize()
        for k in T.SCALARS:
            tk = (2e-2 if "_post_" in k else 5e-3) if dtype == "bf16" else 1e-3
            assert abs(float(out[k]) - float(orc[k])) <= tk * max(abs(float(orc[k])), 1e-6), (dtype, grouped, k, float(out[k]), float(orc[k]))
        sdg = {k.replace(".module.", "."): p for k, p in model.named_parameters()}
        fa, fb, ratios = [], [], {}
        for k in ost:
            if not O.is_param(k) or k.endswith("conv0.bias"):      # conv0.bias: analytically zero gradient (a BatchNorm follows)
                continue
            a_, b_ = sdg[k].grad.detach().double().cpu().flatten(), ost[k].grad.double().flatten()
            fa.append(a_)
            fb.append(b_)
            ratios[k] = float(a_.norm() / b_.norm().clamp_min(1e-30))
        fa, fb = torch.cat(fa), torch.cat(fb)
        cos = float(fa @ fb / fa.norm() / fb.norm())
        grel = float((fa - fb).norm() / fb.norm())
        big = {k: r for k, r in ratios.items() if "weight" in k and "norm" not in k and ".bias" not in k}
        print("\n[conditioned point, %s %s] vs fp32 oracle gradient: flat cosine %.5f, relative L2 %.4f, conv / linear weight norm "
              "ratios %.3f .. %.3f" % (dtype, "grouped" if grouped else "sequential", cos, grel, min(big.values()), max(big.values())))
        if dtype == "fp32":
            assert cos >= 0.9999 and grel <= 1.5e-2, (cos, grel)
        else:
            assert cos >= 0.955 and grel <= 0.30, (grouped, cos, grel)
            assert 0.9 < min(big.values()) and max(big.values()) < 1.1, sorted(big.items(), key=lambda kv: kv[1])[:3]
