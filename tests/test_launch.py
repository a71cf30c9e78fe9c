"""upmix_amd.launch: N processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, joined over upmix_amd.rendezvous;
a failing rank ends the job with its exit code instead of leaving the others waiting."""
import os
import sys

import pytest

from upmix_amd import launch

PROBE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "rank_probe.py")


@pytest.mark.timeout(120)
def test_three_ranks_meet(tmp_path):
    assert launch.run(3, [sys.executable, PROBE, str(tmp_path), "-1"]) == 0
    for r in range(3):
        assert open(tmp_path / f"rank{r}.txt").read() == "3 2.0 2.0"


@pytest.mark.timeout(120)
def test_a_failing_rank_ends_the_job(tmp_path):
    # rank 1 exits with 7; the others lose the group and fail too - whichever the launcher sees first is reported
    assert launch.run(3, [sys.executable, PROBE, str(tmp_path), "1"]) in (7, 1)
    assert not os.path.exists(tmp_path / "rank1.txt")


def test_command_line(tmp_path):
    assert launch.main(["--nproc", "2", PROBE, str(tmp_path), "-1"]) == 0
    assert open(tmp_path / "rank1.txt").read() == "2 1.0 1.0"
    with pytest.raises(SystemExit):
        launch.main(["--nproc", "2"])
