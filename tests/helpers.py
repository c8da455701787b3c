"""Shared test helpers: run one forward/backward of a framework Network and of a torch oracle on the same data."""
import torch
import torch.nn.functional as F

from distributedpytorch_amd.nn.engine import Tape, input_pad, pad_to
from distributedpytorch_amd.nn.loss import Criterion


def to_nhwc_padded(x_nchw, dtype, device):
    n, c, h, w = x_nchw.shape
    out = torch.zeros(n, h, w, input_pad(c, dtype), dtype=dtype, device=device)
    out[..., :c] = x_nchw.permute(0, 2, 3, 1).to(device=device, dtype=dtype)
    return out


def framework_step(net, x_nchw, labels, criterion=None, train=True):
    """Returns (logits[N, classes] fp32 cpu, loss, acc).  Gradients land in net.store.g."""
    store = net.store
    dev = store.device
    x = to_nhwc_padded(x_nchw, store.compute_dtype, dev)
    labels = labels.to(dev)
    net.train() if train else net.eval()
    tape = Tape() if train else None
    ctx, out = net.forward(x, tape)
    crit = criterion or Criterion()
    res = torch.zeros(2, device=dev)
    if isinstance(out, tuple):
        main, aux = out
        crit(ctx, main, labels, net.num_classes, res)
        crit(ctx, aux, labels, net.num_classes, res, scale=0.4, aux=True)
    else:
        main = out
        crit(ctx, main, labels, net.num_classes, res)
    if train:
        tape.backward()
    logits = main.data.reshape(main.data.shape[0], -1)[:, : net.num_classes].float().cpu()
    return logits, float(res[0]), float(res[1])


def oracle_step(model, x_nchw, labels, train=True):
    model.train() if train else model.eval()
    model.zero_grad()
    with torch.set_grad_enabled(train):
        out = model(x_nchw)
        if isinstance(out, tuple):
            loss = F.cross_entropy(out[0], labels) + 0.4 * F.cross_entropy(out[1], labels)
            out = out[0]
        else:
            loss = F.cross_entropy(out, labels)
        if train:
            loss.backward()
    return out.detach(), float(loss)


def grads_by_name(net):
    st = net.store
    return {s.name: st.storage_to_logical(s, st.gview(s)).float().cpu() for s in st.params}
