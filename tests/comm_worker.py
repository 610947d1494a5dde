"""World-size-2 gloo check of fries_amd.comm.TorchComm on CPU tensors (no GPU, no engine): the two collectives are
driven through the C function pointers of the fries_comm struct exactly as libfries_hip.so drives them, and the bytes
are used the way the engine uses them -- rank-ordered sums of gathered doubles (sum_mpi) and the spawn exchange's
[pass 0 | pass 1] segments routed by the proc hash -- then compared with the multi-rank CPU oracle."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    import torch
    import torch.distributed as dist
    from fries_amd.comm import TorchComm, REC_BYTES
    dist.init_process_group("gloo")
    rank, P = dist.get_rank(), dist.get_world_size()
    comm = TorchComm(1000, "cpu")
    st = comm.struct
    # --- all-gather + sum in rank order
    x = np.array([0.1 * (rank + 1), 1e-17 * (rank + 3)])
    comm.small_send[:16] = torch.from_numpy(x.view(np.uint8).copy())
    assert st.allgather(None, 16, None) == 0
    got = comm.small_recv[:16 * P].numpy().view(np.float64).reshape(P, 2)
    g = 0.0
    for p in range(P):
        g += got[p, 0]
    expect = 0.0
    for p in range(P):
        expect += 0.1 * (p + 1)
    assert g == expect and np.array_equal(got[:, 1], [1e-17 * (p + 3) for p in range(P)])
    # --- all-to-all of 16-byte records: rank r sends (r + 1) * (d + 1) records to rank d, tagged (r, d, i)
    send_counts = [(rank + 1) * (d + 1) for d in range(P)]
    recs = []
    for d in range(P):
        for i in range(send_counts[d]):
            recs.append((rank * 1000003 + d * 1009 + i, float(rank) + d / 16.0 + i * 1e-3))
    buf = np.zeros(len(recs), dtype=[("det", "u8"), ("val", "f8")])
    buf["det"] = [a for a, _ in recs]; buf["val"] = [b for _, b in recs]
    raw = torch.from_numpy(buf.view(np.uint8).copy())
    comm.big_send[:raw.numel()] = raw
    sb = (C.c_uint64 * P)(*[REC_BYTES * c for c in send_counts])
    rb = (C.c_uint64 * P)(*[REC_BYTES * (s + 1) * (rank + 1) for s in range(P)])
    assert st.alltoallv(None, sb, rb, None) == 0
    n_recv = sum((s + 1) * (rank + 1) for s in range(P))
    out = comm.big_recv[:REC_BYTES * n_recv].numpy().view([("det", "u8"), ("val", "f8")])
    k = 0
    for s in range(P):                   # source-rank order, each source's records in its own order
        for i in range((s + 1) * (rank + 1)):
            assert out["det"][k] == s * 1000003 + rank * 1009 + i and out["val"][k] == float(s) + rank / 16.0 + i * 1e-3
            k += 1
    assert comm.n_allgather == 1 and comm.n_alltoallv == 1
    dist.barrier()
    dist.destroy_process_group()
    print("comm ok", rank)


if __name__ == "__main__":
    main()
