"""The xGMI gradient exchange (csrc/mopoe_xgmi.inc, comm.py) between real
processes: W ranks, each with its own window, all on the box's one GPU (peer
windows between processes of the same device use the same IPC path as between
the GPUs of a node).  Checked bit for bit against the rank-ordered sum and
against `all-reduce + mopoe_adam_step`."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_exchange_and_fused_adam_between_processes(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0",
                   # W processes share this box's one GPU: the fused launch assumes its
                   # whole grid is resident, which other processes' grids on the same CUs
                   # break.  The library finds that out by itself (mopoe_comm_connect compares
                   # the device UUIDs that travel with the IPC handles) and runs the steps of
                   # this communicator in separate launches -- same bits, tests/test_hip_fused.py;
                   # on the node every rank has its own GPU.  (No MOPOE_NO_FUSE here any more.)
                   # the exchange INSIDE the weight-gradient launch is rehearsed with two
                   # ranks only: its waiting workgroups hold most of a CU's registers,
                   # and with three other ranks' launches waiting on the same GPU a
                   # rank's 1024-thread row-group workgroups may find no CU to start on
                   # (a circular wait over CUs that needs ranks sharing a GPU)
                   XGMI_IN_BACKWARD="1" if world == 2 else "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "xgmi_worker.py")],
                                      env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=240)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    bad = ["rank %d (rc %s):\n%s" % (r, p.returncode, out[-1500:])
           for r, (p, out) in enumerate(zip(procs, outs)) if p.returncode != 0]
    assert not bad, "\n".join(bad)
    if os.environ.get("XGMI_TIME"):
        print("\n".join(l for out in outs for l in out.splitlines() if "us per" in l))
