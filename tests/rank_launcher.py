"""Rank launcher for the multi-process GPU tests.  A process that has initialised the GPU must not fork + exec another program on
this pool, and the pytest process has done so long before tests/test_gpu_multi.py runs.  conftest.py therefore starts THIS helper at
session start, before any test touches the GPU; it never imports torch or opens the device, and starts the rank processes on request.

Protocol (one JSON object per line): stdin {"argv": [...], "env": {...}, "world": N, "timeout": s} -> stdout {"rc": [rc_0 .. rc_N-1]}.
Every rank gets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT like torch.distributed.run provides; its output goes to
this process's stderr."""
import json
import os
import socket
import subprocess
import sys


def main():
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        req = json.loads(line)
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        world = int(req["world"])
        procs = []
        for r in range(world):
            env = dict(os.environ, **req.get("env", {}), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen(req["argv"], env=env, stdout=sys.stderr, stderr=sys.stderr))
        rcs = []
        for p in procs:
            try:
                rcs.append(p.wait(timeout=float(req.get("timeout", 600))))
            except subprocess.TimeoutExpired:
                p.kill()            # the exact child this helper started
                rcs.append(-9)
        sys.stdout.write(json.dumps({"rc": rcs}) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
