"""The sharded (multi-rank) orchestrator on real HIP kernels: two ranks share cuda:0 and talk over gloo
(RCCL refuses two ranks on one device; the collectives used are backend-agnostic).  Checks that the
row-sharded bank + all-gather/merge + replicated label table reproduce the single-process result and the
reference fixture."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, golden_dir, name, ret):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "open-hummingbird-eval_amd"), os.path.join(root, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from helpers import IndexedReplayExtractor, golden_case_indexed
    from hbird_mi.hbird_eval import HbirdEvaluation
    g = np.load(f"{golden_dir}/g67_memory_evaluate.npz")
    c = golden_case_indexed(g, name)
    torch.set_rng_state(torch.from_numpy(g[f"rng_state_{name}"]))
    ev = HbirdEvaluation(IndexedReplayExtractor(c["tokens_by_key"], c["S"], c["D"]), c["train"], num_classes=c["C"],
                         n_neighbours=c["k"], augmentation_epoch=c["aug"], device="cuda:0", nn_method="faiss",
                         nn_params={"idx_shard": True}, memory_size=c["mem"], dataset_size=c["nb"] * c["B"])
    assert ev.sharded and ev.total_rows == g[f"feature_memory_{name}"].shape[0]
    # this rank's rows are a contiguous slice of the reference bank, in the reference's order
    fm = ev.feature_memory.numpy()
    ref = g[f"feature_memory_{name}"][ev.id_base: ev.id_base + fm.shape[0]]
    ok = fm.shape == ref.shape and np.abs(fm - ref).max() <= 2.5e-7
    ok = ok and np.array_equal(ev.label_memory.numpy(), g[f"label_memory_{name}"][ev.id_base: ev.id_base + fm.shape[0]])
    jac, det = ev.evaluate(c["val"], c["S"], return_knn_details=True, ignore_index=c["ign"])
    ok = ok and abs(jac - float(g[f"jac_{name}"])) < 1e-4
    # each rank holds the details of its own validation batches (round-robin): batch `rank`
    B = c["B"]
    lh_ref = g[f"label_hat_{name}"][rank * B:(rank + 1) * B]
    lh = det["knns_ca_labels"].numpy()
    ok = ok and lh.shape == lh_ref.shape and (np.abs(lh - lh_ref) < 5e-5).mean() > 0.999
    rs = g[f"knns_rowsum_{name}"][rank * B:(rank + 1) * B]
    ok = ok and (np.abs(det["knns"].numpy().sum(-1) - rs) < 1e-4).mean() > 0.995
    ret[rank] = (bool(ok), float(jac))
    td.destroy_process_group()


@pytest.mark.parametrize("name", ["unb", "trim"])
def test_two_rank_sharded_evaluation(cuda_device, golden_dir, name):
    world, port = 2, _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, golden_dir, name, ret), nprocs=world, join=True)
    assert ret[0][0] and ret[1][0], dict(ret)
    assert ret[0][1] == ret[1][1]          # every rank reports the same (all-reduced) mIoU
