"""GPU: the device-resident step state (include/vfm_hip.h: vfm_dev_step_t -- step-dependent constants in device memory, so a
captured HIP graph of forward + fused backward replays as consecutive training steps) and the packed first-order records
(vfm_problem_t.wrec), both against the eager step with host-side constants: the trajectory must be the eager one BIT FOR
BIT -- parameters, Adam moments, losses -- across moment-period boundaries and learning-rate changes.
Reference loop: vfm-torch.py:351-370."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(sizes, d, B, nb, output="reg", **attrs):
    from vae_amd.model import VFM
    from vae_amd.data import synthetic_triples
    torch.manual_seed(3)
    m = VFM(field_sizes=list(sizes), embedding_size=d, device="cuda", rng_seed=11, output=output)
    for k, v in attrs.items():
        assert hasattr(m, k), k
        setattr(m, k, v)
    X, y = synthetic_triples(list(sizes), nb * B, seed=4, device="cuda", output=output)
    m.set_training_data(X, nb_train=nb * B)
    plans = [m.plan(X[i * B:(i + 1) * B], y[i * B:(i + 1) * B]) for i in range(nb)]
    return m, plans, X


def test_device_step_state_replays_consecutive_steps_bitwise():
    """The ONE capturability test of the device-resident step state (vfm_dev_step_t, ops.StepState): forward + fused
    backward/Adam captured ONCE with the Philox step and the Adam constants in device memory, replayed as consecutive
    training steps (the kernels advance the counters themselves) == the same steps launched eagerly with host-side
    arguments, bit for bit -- across a learning-rate change (table refill; the error flag survives it) and a moment-period
    boundary.  (`VFM.fit` itself launches eagerly: a graph launch measured slower than two direct launches on ROCm 7.2.)"""
    from vae_amd import ops
    dev = torch.device("cuda:0")
    eager, plans_e, _ = _setup((1500, 500), 64, 200, 1, pipeline=False)
    rep, plans_r, _ = _setup((1500, 500), 64, 200, 1, pipeline=False)
    plan_e, plan_r = plans_e[0], plans_r[0]
    for m in (eager, rep):
        m._ensure_opt_state()
        m._set_moment_form(True)

    def launch(m, plan, bufs, lr, step, dev_step=None):
        sumz, grow, pred, loss3 = bufs
        ent, bia, scal = m._views(m._flat)
        st = ops.elbo_forward(plan, ent, bia, scal, m.inv_occ, seed=11, step=step, train=True, out_pred=pred, out_sumz=sumz,
                              out_grow=grow, out_partials=m._partials, dev_step=dev_step)
        ops.elbo_backward_adam(plan, st, ent, bia, scal, m.inv_occ, m._views(m._adam_m), m._views(m._adam_v), lr, step + 1,
                               loss_out=loss3, scaled_moments=True, dev_step=dev_step)

    def buffers(m):
        return (torch.empty(200, m.d, device=dev), torch.empty(200, device=dev), torch.empty(200, device=dev),
                torch.empty(3, device=dev))
    be, br = buffers(eager), buffers(rep)
    st8 = ops.StepState(dev)
    plan_r.index_tensors()
    launch(rep, plan_r, br, 0.05, 0)                # warm-up outside the capture (lazily made streams, the allocator) ...
    launch(eager, plan_e, be, 0.05, 0)              # ... mirrored on the eager model: both are at step 1 now
    st8.cover(2, 0.05, 0.9, 0.999, 1e-8, True)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            launch(rep, plan_r, br, 0.0, 0, dev_step=st8.dev)      # (host-side lr / step are ignored with dev_step)
    losses_e, losses_r = [], []
    st8.set(1, 2)
    for step in range(1, 140):                      # Adam steps 2 .. 140: across the period boundary at 128
        lr = 0.05 if step < 70 else 0.02
        st8.cover(step + 1, lr, 0.9, 0.999, 1e-8, True)
        st8.set(step, step + 1)
        g.replay()
        st8.advanced()
        launch(eager, plan_e, be, lr, step)
        losses_e.append(be[3].clone()); losses_r.append(br[3].clone())
    torch.cuda.synchronize()
    assert not st8.error()
    assert all(torch.equal(a, b) for a, b in zip(losses_e, losses_r))
    assert torch.equal(eager._flat, rep._flat) and torch.equal(eager._adam_m, rep._adam_m) and torch.equal(eager._adam_v, rep._adam_v)
    # a step the table does not cover raises the sticky error flag -- and a later refill must not erase it
    st8.set(5000, 5000)
    g.replay()
    st8.cover(1, 0.05, 0.9, 0.999, 1e-8, True)
    assert st8.error()


def test_packed_first_order_records_do_not_change_the_trajectory(monkeypatch):
    """use_wrec on / off: the forward reads the same (mu_w, s_w, 1/occ) either way -- bitwise the same run; the records
    survive look-ahead steps, catch-up passes and predictions, and are dropped by an unfused step, a load_state_dict and
    a direct write announced with params_changed()."""
    import vae_amd.model as M
    monkeypatch.setattr(M, "_CHECK_WREC", True)
    a, pa, X = _setup((1500, 500), 64, 200, 5, pipeline=False, use_wrec=True)
    b, pb, _ = _setup((1500, 500), 64, 200, 5, pipeline=False, use_wrec=False)
    for s in range(140):
        fused = s not in (40, 41)
        la_, _ = a.train_step(pa[s % 5], lr=0.04, next_plan=pa[(s + 1) % 5], fused=fused)
        lb_, _ = b.train_step(pb[s % 5], lr=0.04, next_plan=pb[(s + 1) % 5], fused=fused)
        assert a._wrec_ok == fused and b._wrec is None
        if s % 20 == 3:
            assert torch.equal(la_, lb_), s
        if s == 77:
            assert torch.equal(a.predict(X[:64])["y_pred"], b.predict(X[:64])["y_pred"])
    sd = a.state_dict()
    a.sync_lazy(); b.sync_lazy()
    assert torch.equal(a._flat, b._flat)
    assert a._wrec_ok
    a.load_state_dict(sd)
    assert not a._wrec_ok and not a._lazy_dirty
    a.train_step(pa[0], lr=0.04, next_plan=pa[1])
    assert a._wrec_ok
    a.bias_params.weight.data[3, 0] = 0.25
    a.params_changed()
    assert not a._wrec_ok
    a.train_step(pa[1], lr=0.04, next_plan=pa[2])           # (rebuilt from the tables: the debug check holds again)
    assert a._wrec_ok
