#!/usr/bin/env python3
"""Where the tile partition kernel of the CSR pass spends its time: a -DDME_CSR3_TS build of the library
adds thread 0's shader cycles per phase to the first words of the workspace.

  hipcc ... -DDME_CSR3_TS -c dme_csr2.hip ; link as bayespy_amd/libdme_ts3.so   (see benchmarks/README or the Makefile's ts3 target)
  python benchmarks/csr_phases.py --rows 4000000 --lib bayespy_amd/libdme_ts3.so
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "benchmarks"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4_000_000)
    ap.add_argument("--k", type=int, default=50_000)
    ap.add_argument("--lib", required=True)
    args = ap.parse_args()
    from bayespy_amd import _lib
    _lib.LIB_PATH = os.path.abspath(args.lib)
    import torch
    from bow_csr import generate
    L = _lib.lib()
    rowptr, colidx, vals = generate(args.rows, args.k)
    N, nnz, K = rowptr.numel() - 1, colidx.numel(), args.k
    nbins, tbins = int(vals.max()) + 1, 2001
    need = L.dme_stats_csr_workspace_bytes(N, nnz, K)
    ws = torch.zeros(need + 256, dtype=torch.uint8, device="cuda")
    off = (-ws.data_ptr()) % 256
    for rep in range(2):
        hist = torch.zeros(K * nbins, dtype=torch.int64, device="cuda")
        histV = torch.zeros(tbins, dtype=torch.int64, device="cuda")
        acc = torch.zeros(K, dtype=torch.float64, device="cuda")
        status = torch.zeros(8, dtype=torch.int64, device="cuda")
        _lib.check(L.dme_stats_csr_i32(rowptr.data_ptr(), colidx.data_ptr(), vals.data_ptr(), N, nnz, K,
                                       hist.data_ptr(), nbins, histV.data_ptr(), tbins, acc.data_ptr(),
                                       status.data_ptr(), 0, ws.data_ptr() + off, need,
                                       torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
    ts = ws[off:off + 64].view(torch.int64).cpu().tolist()
    tot = sum(ts)
    names = ["P1 window+prefix", "P2 PX/F/rank/loads", "P3 totals/offsets/F scan", "P4 records", "P5 flush",
             "loop head (wait for loads)", "barriers", "-"]
    print("rows", N, "nnz", nnz, "status", status.cpu().tolist())
    for n, t in zip(names, ts):
        print("  %-28s %14d cycles  %5.1f %%" % (n, t, 100.0 * t / max(tot, 1)))


if __name__ == "__main__":
    main()
