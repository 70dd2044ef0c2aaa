"""Probe: HIP streams restricted to a subset of the CUs (hipExtStreamCreateWithCUMask).

Question: the MFMA-bound weight-gradient kernels fill a CU's registers and LDS, so the HBM-bound
chain of the backward never runs beside them on the same CU.  If the weight-gradient stream is confined to N of the 256 CUs, the
chain gets the others: how fast is (a) an HBM-bound kernel on 256-N CUs, (b) the weight gradient on N CUs, (c) both together?"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, ".")
from fusion_gcn_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))


def masked_stream(bits):
    """bits: iterable of CU indices allowed -> torch ExternalStream"""
    words = [0] * 8
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    arr = (C.c_uint32 * 8)(*words)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def t_ms(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


B, V, Cc, T = 128, 25, 128, 150
g = torch.randn(B, T, V, Cc, device=dev)
du = torch.randn(B, T, V, Cc, device=dev)
y = torch.randn(B, T, V, Cc, device=dev)
x = torch.randn(B, T, V, Cc, device=dev)
VEC = torch.cat([torch.zeros(Cc), torch.ones(Cc), torch.ones(Cc), torch.zeros(Cc)]).to(dev).view(4, Cc).contiguous()
ops.set_math_mode("bf16x3")


def mfma():
    ops.tconv_wgrad(g, du, taps=9, stride=1, conv_param=(1, Cc))


def hbm():
    for _ in range(3):
        ops.bn_act(y, VEC, x, None, relu=True)


main = torch.cuda.current_stream()
print(f"full machine: wgrad {t_ms(mfma):.3f} ms, 3x bn_act {t_ms(hbm):.3f} ms (3 x 3 x {y.numel() * 4 / 1e6:.0f} MB)")
for layout in ("low", "spread"):
    for n_side in (64, 128, 192, 224):
        if layout == "low":
            side_bits = list(range(n_side))
        else:                                  # every XCD keeps the same share: CU i of 256 belongs to the side set if i % 8 < n_side / 32
            side_bits = [i for i in range(256) if (i % 32) < n_side // 8]
        rest_bits = [i for i in range(256) if i not in set(side_bits)]
        side, rest = masked_stream(side_bits), masked_stream(rest_bits)

        def on(stream, fn):
            def run():
                stream.wait_stream(main)
                with torch.cuda.stream(stream):
                    fn()
                main.wait_stream(stream)
            return run

        def both():
            side.wait_stream(main)
            rest.wait_stream(main)
            with torch.cuda.stream(side):
                mfma()
            with torch.cuda.stream(rest):
                hbm()
            main.wait_stream(side)
            main.wait_stream(rest)

        def both_unmasked_main():
            side.wait_stream(main)
            with torch.cuda.stream(side):
                mfma()
            hbm()
            main.wait_stream(side)
        a, b, ab, ab2 = t_ms(on(side, mfma)), t_ms(on(rest, hbm)), t_ms(both), t_ms(both_unmasked_main)
        print(f"[{layout}] wgrad on {len(side_bits)} CUs {a:.3f} ms | 3x bn_act on {len(rest_bits)} CUs {b:.3f} ms | together {ab:.3f} ms | "
              f"wgrad masked + bn_act unmasked {ab2:.3f} ms")
