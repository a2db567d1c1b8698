"""The `main` command line keeps the reference's contract (main.cpp:43-128)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN = os.path.join(ROOT, "htm-hashjoin_amd", "bin", "main")


def run(*args):
    return subprocess.run([MAIN, *map(str, args)], capture_output=True, text=True)


def test_unknown_arg_prints_and_exits_1():          # main.cpp:64-65
    r = run("--bogus", 1)
    assert r.returncode == 1 and r.stdout.strip() == "Found Unknown Arg: --bogus"


def test_unknown_algo_prints_and_exits_0():         # main.cpp:108
    r = run("--algo", "quantum", "--rSize", 1024)
    assert r.returncode == 0 and r.stdout.strip() == "Unknown Algo: quantum"


def test_unknown_distribution():                    # DataGen.hpp:116-119
    r = run("--algo", "nocc", "--rSize", 1024, "--dataDistr", "gauss")
    assert r.returncode == 1 and "Unknown distribution" in r.stdout


def test_config1_plumbing_nocc_uniform():
    """BASELINE.json configs[0]: --algo nocc --rSize 1048576 --dataDistr uniform on the CPU path."""
    r = run("--algo", "nocc", "--rSize", 1048576, "--dataDistr", "uniform", "--numPartitions", 1)
    assert r.returncode == 0
    j = json.loads(r.stdout)
    # field order of NoCCHashBuild.hpp:127-146
    assert list(j)[:8] == ["algo", "rSize", "probeLength", "hashBuildTimeInMicroseconds", "conflicts",
                           "totalMatches", "inputSum", "outputSum"]
    # one partition = sequential order = the pinned values
    assert (j["conflicts"], j["totalMatches"], j["inputSum"], j["outputSum"]) == (
        176864, 871712, 549507039110, 549504941959)


@pytest.mark.parametrize("algo,outsum", [("nocc", 549755289600), ("cpu-atomic", 549756338176)])
def test_unique_keys_any_thread_count(algo, outsum):
    r = run("--algo", algo, "--rSize", 1048576, "--probeLength", 4, "--dataDistr", "local_shuffle",
            "--shuffleRange", 1024)
    j = json.loads(r.stdout)
    assert (j["conflicts"], j["totalMatches"], j["inputSum"], j["outputSum"]) == (0, 1048576, 549756338176, outsum)


def test_build_only_variant_has_no_totalMatches():  # ENABLE_PROBE 0, main.cpp:113-120
    j = json.loads(run("--algo", "nocc", "--rSize", 4096, "--dataDistr", "sorted", "--probe", 0).stdout)
    assert "totalMatches" not in j and j["conflicts"] == 0


def test_gpu_algos_do_not_fall_back_to_cpu():
    import htm_hashjoin_amd as hj
    if hj.device_count() > 0:
        pytest.skip("GPU present")
    for algo in ("atomic", "htm", "prj"):
        r = run("--algo", algo, "--rSize", 1024, "--dataDistr", "sorted")
        assert r.returncode == 2 and "no gfx950" in r.stderr and r.stdout == ""
