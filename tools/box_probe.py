"""The box's ceilings (tools/micro/box_probe.hip): streaming-read GB/s over two 65.5 MB buffers (the bytes of one headline launch) and the
integer VALU issue rate.  Usage: python tools/box_probe.py"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load():
    lib = C.CDLL(os.path.join(ROOT, "tools", "micro", "libbox_probe.so"))
    lib.box_stream_read_ms.restype = C.c_float
    lib.box_stream_read_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.box_valu_issue.restype = C.c_double
    lib.box_valu_issue.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p]
    return lib


def main():
    import torch
    lib = load()
    dev = torch.device("cuda", 0)
    n = 16384 * 1000
    out = {"stream": [], "valu": []}
    # several pairs of buffers, cycled by the caller, would defeat the 256 MB Infinity Cache; one pair of 65.5 MB each fits it — so both are measured
    a = torch.randint(1, 1000, (n,), dtype=torch.int32, device=dev)
    b = torch.randint(1, 1000, (n,), dtype=torch.int32, device=dev)
    big_a = torch.randint(1, 1000, (8 * n,), dtype=torch.int32, device=dev)
    big_b = torch.randint(1, 1000, (8 * n,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for gpc in (2, 4, 8, 16):
        for nt in (0, 1):
            ms = lib.box_stream_read_ms(a.data_ptr(), b.data_ptr(), n * 4, 20, gpc, nt, None)
            ms_big = lib.box_stream_read_ms(big_a.data_ptr(), big_b.data_ptr(), 8 * n * 4, 5, gpc, nt, None)
            out["stream"].append({"wg_per_cu": gpc, "nt": nt, "us_131MB_same_buffers": round(ms * 1e3, 2), "gbs_same": round(2 * n * 4 / ms / 1e6, 1),
                                  "us_per_131MB_of_1GB": round(ms_big * 1e3 / 8, 2), "gbs_1GB": round(2 * 8 * n * 4 / ms_big / 1e6, 1)})
    for pk in (0, 1):
        for wps in (1, 2, 4):
            cyc, mhz = C.c_double(), C.c_double()
            r = lib.box_valu_issue(pk, wps, C.byref(cyc), C.byref(mhz), None)
            out["valu"].append({"packed": pk, "waves_per_simd": wps, "wave_inst_per_s_per_cu": r, "simd_cycles_per_inst_at_reported_clock": round(cyc.value, 3), "mhz": mhz.value})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
