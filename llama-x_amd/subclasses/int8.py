"""INT8 row-wise quantised linear weight as a tensor subclass, HIP-backed.

Public surface and behaviour follow /root/reference/subclasses/int8.py:10-130 (quantiser :10-16, wrapper subclass
:19-102, autograd function :106-130); the device arithmetic is in llama-x_amd/csrc (int8_quant.hip, gemm_bf16.hip).
"""
import torch
import torch.nn.functional as F
from torch import Tensor

from llx import _lib as L
from llx import kernels as K

from .int8_mm import _launch as _i8_gemm
from .int8_mm import int8_mm_dequant

aten = torch.ops.aten


def quantize_int8_rowwise(x: Tensor):
    """absmax/127 per row in fp32, divide by clip(scale, 1e-12), round half-to-even, int8; scale in x's dtype."""
    if x.is_cuda:
        assert x.dim() == 2
        if x.dtype not in (torch.bfloat16, torch.float32):
            raise L.LlxError(f"quantize_int8_rowwise: unsupported dtype {x.dtype} on the HIP path")
        xc = x if x.stride(1) == 1 else x.contiguous()
        q = torch.empty(xc.shape, device=x.device, dtype=torch.int8)
        s = torch.empty(xc.shape[0], device=x.device, dtype=x.dtype)
        L.check(L.load().llx_quantize_int8_rowwise(L.ptr(xc), xc.stride(0), L.ptr(q), q.stride(0), L.ptr(s), xc.shape[0], xc.shape[1],
                                                   int(x.dtype is torch.float32), L.stream()), "llx_quantize_int8_rowwise")
        return q, s
    # host-side (one-off: the scripts quantise the model before .cuda(), train_metamathqa.py:178,184)
    xf = x.to(torch.float32)
    scale = xf.abs().amax(dim=1) / 127
    q = torch.round(xf / scale.clamp_min(1e-12).unsqueeze(1)).to(torch.int8)
    return q, scale.to(x.dtype)


def _bf16_image(w: "Int8LinearWeight") -> Tensor:
    """bf16 copy of the int8 matrix (int8 is exact in bf16) - B operand of the weight-only GEMM."""
    from llx.ops import _cached

    return _cached(w.int_data, "bf16", lambda: K.i8_to_bf16(w.int_data))


def int8_weight_t(w: "Int8LinearWeight") -> Tensor:
    """[K,N] bf16 transpose of the int8 matrix - B operand of the backward GEMM (subclasses/int8.py:127)."""
    from llx.ops import _cached

    return _cached(w.int_data, "wt", lambda: K.transpose(w.int_data))


def int8_linear_forward(x2: Tensor, w: "Int8LinearWeight", out: Tensor | None = None) -> Tensor:
    """_Int8Linear.forward without bias on [M,K] rows (subclasses/int8.py:106-121)."""
    if w.dynamic_int8_act:
        xi, xs = quantize_int8_rowwise(x2)
        return _i8_gemm(xi, w.int_data, xs, w.scale, out=out)
    return K.gemm_nt(x2, _bf16_image(w), out=out, epilogue=K.EPI_COLSCALE, e=w.scale)


class Int8LinearWeight(Tensor):
    @staticmethod
    @torch._dynamo.disable
    def __new__(cls, int_data: Tensor, scale: Tensor, dynamic_int8_act: bool = False):
        # a wrapper subclass: reports the int8 matrix's shape and the SCALE's dtype (modelling/lora.py relies on that)
        return Tensor._make_wrapper_subclass(cls, int_data.shape, dtype=scale.dtype, device=int_data.device)

    @torch._dynamo.disable
    def __init__(self, int_data: Tensor, scale: Tensor, dynamic_int8_act: bool = False):
        assert int_data.dtype is torch.int8
        assert int_data.ndim == 2
        assert scale.ndim == 1
        self.int_data = int_data
        self.scale = scale
        self.dynamic_int8_act = dynamic_int8_act

    # ---- serialisation hooks (state_dict / torch.save round trip, train_librispeech.py:200-204)
    def __tensor_flatten__(self):
        return ["int_data", "scale"], [self.dynamic_int8_act]

    @classmethod
    def __tensor_unflatten__(cls, tensor_data_dict, tensor_attributes, outer_size=None, outer_stride=None):
        return cls(tensor_data_dict["int_data"], tensor_data_dict["scale"], *tensor_attributes)

    @classmethod
    def from_float(cls, tensor: Tensor, dynamic_int8_act: bool = False):
        int_data, scale = quantize_int8_rowwise(tensor)
        return cls(int_data, scale, dynamic_int8_act)

    def dequantize(self):
        return self.int_data * self.scale.view(-1, 1)

    def __repr__(self):
        return (f"{self.__class__.__name__}(shape={tuple(self.shape)}, dynamic_int8_act={self.dynamic_int8_act}, "
                f"dtype={self.dtype}, device={self.device}, requires_grad={self.requires_grad})")

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is F.linear:
            return _Int8Linear.apply(*args, **kwargs)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)

    @classmethod
    def __torch_dispatch__(cls, func, types, args, kwargs):
        handler = _DISPATCH.get(func)
        if handler is None:
            raise NotImplementedError(f"{cls.__name__} dispatch: attempting to run {func}, this is not supported")
        return handler(cls, func, args, kwargs)


def _rewrap(cls, func, args, kwargs):
    w = args[0]
    return cls(func(w.int_data, *args[1:], **kwargs), func(w.scale, *args[1:], **kwargs), w.dynamic_int8_act)


def _to_copy(cls, func, args, kwargs):
    w = args[0]
    device, dtype = kwargs.get("device", None), kwargs.get("dtype", None)
    return cls(w.int_data.to(device=device), w.scale.to(device=device, dtype=dtype), w.dynamic_int8_act)


def _copy_(cls, func, args, kwargs):
    dst, src = args[0], args[1]
    if isinstance(dst, cls) and isinstance(src, cls):
        dst.int_data.copy_(src.int_data)
        dst.scale.copy_(src.scale)
    elif isinstance(dst, cls):  # float -> int8: re-quantise
        q, s = quantize_int8_rowwise(src)
        dst.int_data.copy_(q)
        dst.scale.copy_(s)
    else:  # int8 -> float
        dst.copy_(src.dequantize())
    return dst


_DISPATCH = {
    aten.detach.default: _rewrap,
    aten.clone.default: _rewrap,
    aten._to_copy.default: _to_copy,
    aten.copy_.default: _copy_,
}


class _Int8Linear(torch.autograd.Function):
    """F.linear on an Int8LinearWeight (subclasses/int8.py:106-130): no weight gradient, bf16 backward."""

    @staticmethod
    def forward(ctx, input: Tensor, weight: Int8LinearWeight, bias: Tensor | None = None):
        L.require_cuda(input)
        ctx.weight = weight
        x2 = K._rows2d(input)
        out = int8_linear_forward(x2, weight).view(*input.shape[:-1], -1)
        if bias is not None:
            out = out + bias
        return out

    @staticmethod
    def backward(ctx, grad_output: Tensor):
        w: Int8LinearWeight = ctx.weight
        g2 = K._rows2d(grad_output)
        grad_input = None
        if ctx.needs_input_grad[0]:
            grad_input = K.gemm_nt(K.scale(g2, colscale=w.scale), int8_weight_t(w)).view(*grad_output.shape[:-1], w.shape[1])
        grad_bias = None
        if len(ctx.needs_input_grad) > 2 and ctx.needs_input_grad[2]:
            grad_bias = grad_output.reshape(-1, w.shape[0]).sum(0)
        return grad_input, None, grad_bias
