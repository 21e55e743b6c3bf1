"""`build_model` and `ModelWrapper` (torchdet3d/builders/model_builder.py:25-151) over the HIP engine.

Same call surface as the reference: `build_model(config, export_mode=False, weights_path='')` returns an
`nn.Module` whose `forward(x, cats) -> (kp [B,9,2], targets [B,num_classes] | cats[:,None])`, with
`train()/eval()/to()/parameters()/state_dict()/load_state_dict()` (state-dict keys and shapes are the
reference's).  Differences a caller can observe:
  * `parameters()` yields ONE flat fp32 tensor holding every weight (the named tensors are views into it), so an
    optimizer built by `build_optimizer` is a single fused update and data-parallel training all-reduces one
    buffer; `named_parameters()` therefore has one entry, `state_dict()` keeps the per-layer names;
  * the forward runs on the GPU only ('--device cuda'); there is no CPU fallback;
  * extra model names: 'mobilenetv2' (north-star throughput model).  The timm / efficientnet-lite names of the
    reference need un-vendored third-party packages and are not built.
Config keys read: model.name, model.num_classes, model.pretrained (ignored: no network), model.load_weights,
model.storage_dtype ('f32' default for parity | 'bf16' throughput mode)."""
import torch
from torch import nn

from ..models.arch import AVAILABLE_MODELS
from ..models.engine import Net
from ..utils.utils import load_pretrained_weights


class _Run(torch.autograd.Function):
    """The whole network as one autograd node: forward = engine.forward, backward = engine.backward."""

    @staticmethod
    def forward(ctx, flat, wrapper, x, cats, mask, train):
        net = wrapper.net
        kp, logits = net.forward(x, cats, train=train, dropout_mask=mask)
        ctx.wrapper, ctx.train = wrapper, train
        if logits is None:
            logits = torch.empty(0, device=kp.device)
            ctx.mark_non_differentiable(logits)
        return kp, logits

    @staticmethod
    def backward(ctx, dkp, dlogits):
        net = ctx.wrapper.net
        if not ctx.train:
            raise RuntimeError('backward through an eval-mode forward is not supported (BatchNorm uses running stats)')
        if dkp is None:
            dkp = torch.zeros(net.saved['B'], 9, 2, device=net.device)
        if net.num_classes > 1 and dlogits is None:
            dlogits = torch.zeros(net.saved['B'], net.num_classes, device=net.device)
        sync = ctx.wrapper.grad_sync
        if sync is not None:
            sync.start()
        net.backward(dkp.contiguous(), dlogits.contiguous() if net.num_classes > 1 else None)
        if sync is not None:
            sync.finish()
        return net.gflat, None, None, None, None, None


class ModelWrapper(nn.Module):
    def __init__(self, name, num_classes=9, export_mode=False, storage_dtype='f32', device='cpu'):
        super().__init__()
        assert name in AVAILABLE_MODELS, f'Wrong model name parameter. Expected one of {AVAILABLE_MODELS}'
        self.name, self.num_classes, self.export_mode = name, num_classes, export_mode
        self.storage_dtype = torch.bfloat16 if storage_dtype in ('bf16', torch.bfloat16) else torch.float32
        self.grad_sync = None          # optional torchdet3d.parallel.GradSync (one process per GPU)
        self._make(torch.device(device))

    def _make(self, device, state=None):
        self.net = Net(self.name, self.num_classes, device, self.storage_dtype)
        if state is not None:
            self.net.load_state_dict(state)
        self.flat = nn.Parameter(self.net.flat)     # shares storage with the engine's master weights
        if self.grad_sync is not None:
            self.attach_grad_sync(self.grad_sync)

    def attach_grad_sync(self, sync_cls_or_obj):
        from ..parallel import GradSync
        self.grad_sync = GradSync(self.net.gflat) if sync_cls_or_obj is True else sync_cls_or_obj
        self.grad_sync.g = self.net.gflat
        self.net.grad_hook = self.grad_sync.ready
        self.grad_sync.broadcast([self.net.flat] + list(self.net.buffers.values()))

    # ---- nn.Module surface -------------------------------------------------------------------------
    def _apply(self, fn, recurse=True):
        probe = fn(torch.empty(0, device=self.net.device))
        if probe.device != self.net.device:
            self._make(probe.device, self.net.state_dict())
        return self

    def state_dict(self, *args, destination=None, prefix='', keep_vars=False):
        sd = self.net.state_dict()
        out = destination if destination is not None else {}
        for k, v in sd.items():
            out[prefix + k] = v
        return out

    def load_state_dict(self, state_dict, strict=True):
        self.net.load_state_dict(state_dict, strict=strict)
        return self

    def forward(self, x, cats=None, dropout_mask=None):
        if self.export_mode:
            return self.forward_to_onnx(x)
        if not x.is_cuda:
            raise RuntimeError('the HIP path needs the model and the crops on the GPU (no CPU fallback)')
        # train-mode BatchNorm / dropout only when a backward can follow (grad mode is off inside Function.forward)
        train = self.training and torch.is_grad_enabled()
        kp, logits = _Run.apply(self.flat, self, x.float(), cats, dropout_mask, train)
        targets = logits if self.num_classes > 1 else cats.unsqueeze(1)     # model_builder.py:141-144
        return kp, targets

    @torch.no_grad()
    def forward_to_onnx(self, x):
        """All 9 heads (model_builder.py:112-124): kp [9,B,9,2] (sigmoid), class logits (or zeros(B))."""
        outs, logits = [], None
        for k in range(9):
            kp, lg = self.net.forward(x.float(), torch.full((x.shape[0],), k, dtype=torch.int64, device=x.device),
                                      train=False)
            outs.append(kp.clone().view(1, x.shape[0], 9, 2))
            logits = lg
        return torch.cat(outs), (logits if self.num_classes > 1 else torch.zeros(x.shape[0], device=x.device))


def build_model(config, export_mode=False, weights_path=''):
    name = config.model.name
    assert name in AVAILABLE_MODELS, f'Wrong model name parameter. Expected one of {AVAILABLE_MODELS}'
    model = ModelWrapper(name, config.model.num_classes or 9, export_mode, config.model.storage_dtype or 'f32')
    weights = config.model.load_weights or weights_path
    if weights:
        load_pretrained_weights(model, weights)
    return model
