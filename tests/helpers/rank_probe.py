"""Started by tests/test_launch.py under upmix_amd.launch: joins the process group from the environment, reports."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from upmix_amd.rendezvous import Rendezvous   # noqa: E402

out_dir, fail_rank = sys.argv[1], int(sys.argv[2])
with Rendezvous.from_env(timeout=60) as g:
    if g.rank == fail_rank:
        sys.exit(7)
    top = g.allreduce_max([float(g.rank), float(os.environ["LOCAL_RANK"])])
    open(os.path.join(out_dir, f"rank{g.rank}.txt"), "w").write(f"{g.world} {top[0]} {top[1]}")
