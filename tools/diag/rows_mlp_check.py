"""each product of csrc/rows_mlp.hip at a training step's size against torch in float64 (debugging aid)"""
import sys, torch
sys.path.insert(0, "/root/repo")
import spcl_amd
from spcl_amd import native as _n
torch.manual_seed(0)
f32 = _n.dtype_code(torch.float32)
for (M, N, K) in [(188160, 256, 64), (6000, 256, 256), (188160, 64, 256), (50000, 256, 64), (188160, 256, 32), (200000, 256, 16)]:
    g = torch.randn(M, N, device="cuda")
    W = torch.randn(N, K, device="cuda") * 0.1
    x = torch.randn(M, K, device="cuda")
    dx = torch.empty(M, K, device="cuda")
    _n.call("spcl_rows_linear_backward_input", _n.ptr(g), f32, _n.ptr(W), None, M, N, K, _n.ptr(dx), f32, K, _n.stream())
    ref = g.double() @ W.double()
    e1 = float((dx.double() - ref).abs().max() / ref.abs().max())
    ws = torch.empty(_n.call("spcl_rows_linear_backward_weight_workspace_bytes", M, N, K) // 4 + 1, device="cuda")
    dW, db = torch.empty(N, K, device="cuda"), torch.empty(N, device="cuda")
    _n.call("spcl_rows_linear_backward_weight", _n.ptr(g), f32, _n.ptr(x), f32, K, 0, M, N, K, _n.ptr(ws), ws.numel() * 4, _n.ptr(dW),
            _n.ptr(db), _n.stream())
    rW, rb = g.double().t() @ x.double(), g.double().sum(0)
    e2 = float((dW.double() - rW).abs().max() / rW.abs().max())
    e3 = float((db.double() - rb).abs().max() / rb.abs().max())
    y = torch.empty(M, N, device="cuda")
    b = torch.randn(N, device="cuda")
    _n.call("spcl_rows_linear_forward", _n.ptr(x), f32, K, 0, _n.ptr(W), _n.ptr(b), M, K, N, _n.ptr(y), _n.stream())
    ry = x.double() @ W.double().t() + b.double()
    e0 = float((y.double() - ry).abs().max() / ry.abs().max())
    print((M, N, K), f"fwd {e0:.2e}  dx {e1:.2e}  dW {e2:.2e}  db {e3:.2e}")
