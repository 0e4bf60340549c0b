"""The reference's own optimiser loop on the HIP module, verbatim (train.py:34-35, 200-204):

    optimizer = torch.optim.Adam(filter(lambda p: p.requires_grad, RSVQA.parameters()), lr=...)
    criterion = torch.nn.CrossEntropyLoss()
    pred = RSVQA(image, question); loss = criterion(pred, answer)
    optimizer.zero_grad(); loss.backward(); optimizer.step()

against the oracle's autograd + Adam, in fp32 and in bf16 storage (where the MFMA products read bf16 copies of
the weights that must follow every torch-side update), with dropout on.
"""
import pytest
import torch

from oracle import vqa_ref
from test_gpu_model import MID, batch, build, rel_err

pytestmark = pytest.mark.gpu


def _reference_loop(m, batches, lr):
    """train.py:34-35, 200-204, line for line.  Returns the per-step losses."""
    RSVQA = m
    optimizer = torch.optim.Adam(filter(lambda p: p.requires_grad, RSVQA.parameters()), lr=lr)
    criterion = torch.nn.CrossEntropyLoss()
    losses = []
    RSVQA.train()
    for image, question, answer in batches:
        pred = RSVQA(image, question)
        loss = criterion(pred, answer)
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        losses.append(float(loss.detach()))
    return losses


def _batches(kw, dtype, n, B=24, R=25, L=14):
    out, host = [], []
    for step in range(1, n + 1):
        b = batch(kw, B, R, L, seed=100 + step)
        host.append(b)
        out.append((b['feats'].cuda().to(dtype), b['tokens'].cuda(), b['answers'].cuda()))
    return out, host


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_reference_optimizer_loop_matches_oracle(dtype):
    kw, lr, steps = MID, 1e-2, 4
    m, p = build(kw, 1, False, dtype)
    dev_batches, host_batches = _batches(kw, dtype, steps)
    losses = _reference_loop(m, dev_batches, lr)

    # oracle: fp32 master weights, Adam on them; in bf16 storage the step's arithmetic sees the weights (and inputs)
    # rounded to bf16, as the kernels do
    po = {k: v.clone() for k, v in p.items()}
    names = vqa_ref.trainable_names(po)
    mom = {k: torch.zeros_like(po[k]) for k in names}
    var = {k: torch.zeros_like(po[k]) for k in names}
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    strong = {k: torch.ones_like(po[k], dtype=torch.bool) for k in names}    # gradient well above bf16 noise in EVERY step
    for step, b in enumerate(host_batches, 1):
        seen = po if dtype == torch.float32 else {k: v.to(dtype).float() for k, v in po.items()}
        seeds = [vqa_ref.site_seed(1234, step, s) for s in range(4)]
        _, loss_ref, grads = vqa_ref.loss_and_grads(seen, b['feats'].to(dtype).float(), b['tokens'], b['answers'], None, 1,
                                                    drop_seeds=seeds)
        # THE stale-weight detector: a forward on stale weights (bf16 copies that did not follow optimizer.step())
        # would miss these from step 2 on -- at lr = 1e-2 the weights move by a seventh of their range per step
        assert abs(losses[step - 1] - float(loss_ref)) <= 4 * tol * max(1.0, abs(float(loss_ref))), (step, losses, float(loss_ref))
        for k in names:
            strong[k] &= grads[k].abs() >= 0.2 * grads[k].abs().max()
            vqa_ref.adam_step(po[k], grads[k], mom[k], var[k], step, lr)
    sd = m.state_dict()
    for k in names:
        if k.endswith('linear_a.bias'):
            continue     # zero-mean rounding noise as gradient: Adam turns its sign into +-lr steps
        if dtype == torch.float32:
            assert rel_err(sd[k], po[k]) <= 1e-3, k      # (Adam divides by sqrt(v): gradient noise of 1e-6 is amplified where |g| is tiny)
        else:
            # bf16 gradients carry 1e-2 relative noise and Adam normalises the step length: elements whose
            # gradient is noise-sized may step the other way; the bulk of the update must agree
            upd, upd_ref = (sd[k].cpu() - p[k]).double(), (po[k] - p[k]).double()
            cos = float((upd * upd_ref).sum() / (upd.norm() * upd_ref.norm()).clamp_min(1e-30))
            assert cos >= 0.98, (k, cos)
            # ... and element by element where the oracle's gradient stood at >= 20 % of the tensor's largest in all
            # four steps (bf16 noise is then <= 5 % of the element's gradient, and Adam's step m / sqrt(v) moves by
            # about as much): the accumulated update, at most lr * steps long, agrees to a fifth of that length
            sel = strong[k]
            if bool(sel.any()):
                worst = float((upd - upd_ref)[sel].abs().max())
                assert worst <= 0.2 * lr * steps, (k, worst, int(sel.sum()))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_forward_after_torch_optimizer_step_uses_the_new_weights(dtype):
    """After optimizer.step() (a torch-side in-place update of the parameters) the next forward must compute with
    the updated weights: its logits equal, bit for bit, those of a fresh module loaded from the state_dict."""
    kw = MID
    m, _ = build(kw, 2, False, dtype)
    dev_batches, _ = _batches(kw, dtype, 3)
    _reference_loop(m, dev_batches, 1e-2)
    fresh, _ = build(kw, 2, False, dtype, seed=11)
    fresh.load_state_dict(m.state_dict())
    m.eval(), fresh.eval()
    image, question, _ = dev_batches[0]
    with torch.no_grad():
        a, b = m(image, question), fresh(image, question)
        assert torch.equal(a, b)
        # ... and so after p.add_() / p.copy_() on single parameters
        m.linear_classif2.weight.mul_(1.5)
        m.attention[0].linear_va.weight.add_(0.01)
        m.linear_q.bias.copy_(torch.zeros_like(m.linear_q.bias))
        fresh.load_state_dict(m.state_dict())
        a2, b2 = m(image, question), fresh(image, question)
        assert torch.equal(a2, b2) and not torch.equal(a, a2)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_reference_loop_equals_fused_train_step(dtype):
    """The fused path (train_step: forward + CE + backward + Adam in the library) and the reference's loop through
    torch's CrossEntropyLoss / autograd / Adam are the same procedure on the same dropout stream."""
    kw, lr, steps = MID, 1e-3, 3
    ma, _ = build(kw, 1, False, dtype)
    mb, _ = build(kw, 1, False, dtype)
    dev_batches, _ = _batches(kw, dtype, steps)
    la = _reference_loop(ma, dev_batches, lr)
    mb.train()
    lb = [float(mb.train_step(f, q, a, lr=lr)) for f, q, a in dev_batches]
    sa, sb = ma.state_dict(), mb.state_dict()
    init = build(kw, 1, False, dtype)[0].state_dict()
    for x, y in zip(la, lb):
        assert abs(x - y) <= (1e-6 if dtype == torch.float32 else 2e-3) * max(1.0, abs(y)), (la, lb)
    for k, _ in mb._trainable_order():
        if k.endswith('linear_a.bias'):
            continue
        if dtype == torch.float32:
            assert rel_err(sa[k], sb[k]) <= 1e-5, k
        else:
            # one ulp of difference between torch's Adam and the fused kernel can round a bf16 weight copy the
            # other way, after which the gradients differ in the last bf16 digit: compare the updates
            upd_a, upd_b = (sa[k] - init[k]).double(), (sb[k] - init[k]).double()
            cos = float((upd_a * upd_b).sum() / (upd_a.norm() * upd_b.norm()).clamp_min(1e-30))
            assert cos >= 0.995, (k, cos)


def test_second_forward_before_backward_raises():
    """The module keeps the activations of its latest forward only: a backward that belongs to an earlier forward
    raises instead of returning the gradients of the wrong batch."""
    kw = MID
    m, _ = build(kw, 1, False, torch.float32)
    m.train()
    (f1, q1, a1), (f2, q2, a2) = _batches(kw, torch.float32, 2)[0]
    crit = torch.nn.CrossEntropyLoss()
    l1 = crit(m(f1, q1), a1)
    l2 = crit(m(f2, q2), a2)
    with pytest.raises(RuntimeError, match='latest forward'):
        (l1 + l2).backward()
    # an evaluation forward in between invalidates the pending node too
    l1 = crit(m(f1, q1), a1)
    m.eval()
    with torch.no_grad():
        m(f2, q2)
    m.train()
    with pytest.raises(RuntimeError, match='latest forward'):
        l1.backward()
    # the normal order keeps working afterwards
    m.zero_grad()
    l3 = crit(m(f1, q1), a1)
    l3.backward()
    assert m.linear_classif2.weight.grad is not None


def test_adam_moments_survive_a_module_move():
    """model.to() / .cuda() after training has started rebuilds the flat parameter storage; the Adam moments of the
    fused path are carried over by parameter name."""
    kw = MID
    ma, _ = build(kw, 1, False, torch.float32)
    mb, _ = build(kw, 1, False, torch.float32)
    ma.train(), mb.train()
    dev_batches, _ = _batches(kw, torch.float32, 3)
    for i, (f, q, a) in enumerate(dev_batches):
        ma.train_step(f, q, a, lr=1e-3)
        mb.train_step(f, q, a, lr=1e-3)
        if i == 1:
            mb = mb.to('cuda')                 # _apply: the flat buffers are rebuilt on the next call
            mb = mb.float()
    sa, sb = ma.state_dict(), mb.state_dict()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
