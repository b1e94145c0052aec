"""Helpers for the -m gpu parity tests: build single-layer plans through the C-ABI and compare with the oracle."""
import numpy as np
import torch

from segmentation_amd import _lib as L
from segmentation_amd import engine as E


def dev():
    return torch.device('cuda', 0)


def round_dtype(a, dtype):
    """value grid of the compute dtype, returned as float64 for the oracle"""
    a = np.asarray(a, np.float32)
    if dtype == L.SEG_BF16:
        return torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy().astype(np.float64)
    return a.astype(np.float64)


def fill_act(act, arr):
    """arr: logical [B,H,W,C] -> writes channels [0,C), pad channels zero"""
    t = torch.zeros(act.t.shape, dtype=torch.float32)
    t[..., :arr.shape[-1]] = torch.from_numpy(np.asarray(arr, np.float32))
    act.t.copy_(t.to(act.t.dtype))


def read_act(act, C=None):
    C = act.C if C is None else C
    return act.t[..., :C].to(torch.float32).cpu().numpy().astype(np.float64)


def pad_channels_zero(act):
    if act.Cp == act.C:
        return True
    return bool((act.t[..., act.C:].to(torch.float32) == 0).all().item())


def stream():
    return torch.cuda.current_stream().cuda_stream


def sync():
    torch.cuda.synchronize()


def tol(dtype, f32_tol=2e-5, bf16_tol=2e-2):
    return f32_tol if dtype == L.SEG_F32 else bf16_tol


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def make_store(layers, dtype, params):
    store = E.ParamStore(layers, dtype, dev(), training=True)
    store.set_params(params)
    net = E.Net(store, 1, dtype, dev())
    p = E.Plan('pack'); net.pack(p); p.run(stream()); sync()
    return store
