"""How exactly does v_mfma_scale_f32_16x16x128_f8f6f4 accumulate?  The fp8 x fp8 GEMM (float32 output) against float64 on the same
quantised operands, next to the bf16 MFMA GEMM on the SAME values (every e4m3 * 2^e value is a bf16 value, so the products are exact
in both)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import lib
from oracle import model_ref as MR
from tests.util import tile8
L = lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
for (M, N, K) in [(512, 512, 2048), (256, 256, 256), (512, 256, 11008)]:
    x = torch.randn(M, K).to(torch.bfloat16)
    w = (torch.randn(N, K) * 0.04).to(torch.bfloat16)
    QW = MR.QuantW(w.float())
    xq, e = MR.mx_quantize(x.float())
    ref = xq.double() @ QW.q.double().t()
    rp = (M + 255) // 256 * 256
    q = torch.zeros(M, K, dtype=torch.uint8, device="cuda")
    sc = torch.zeros(K // 128, rp, 4, dtype=torch.uint8, device="cuda")
    L.sr_op_quant_mx(P(x.cuda()), K, M, K, P(q), P(sc), rp, s)
    out = torch.zeros(M, N, dtype=torch.float32, device="cuda")
    ones = torch.ones(N, dtype=torch.float32, device="cuda")
    assert L.sr_op_gemm_mx(P(q), K, P(sc), rp, P(tile8(QW.q8.view(torch.uint8)).cuda()), P(ones), M, N, K, P(out), N, None, None, 4, s) == 0
    out2 = torch.zeros(M, N, dtype=torch.float32, device="cuda")
    assert L.sr_op_gemm(P(xq.to(torch.bfloat16).cuda()), K, P(QW.q.to(torch.bfloat16).cuda()), M, N, K, P(out2), N, None, None, None, 4 | 0x200, s) == 0
    torch.cuda.synchronize()
    for name, o in (("mx fp8 mfma", out), ("bf16 mfma on the same values", out2)):
        d = (o.cpu().double() - ref)
        print(f"{M}x{N}x{K} {name}: max abs {float(d.abs().max()):.3e}  rms {float(d.pow(2).mean().sqrt()):.3e}  mean {float(d.mean()):+.3e}  |ref| rms {float(ref.pow(2).mean().sqrt()):.3f}")
