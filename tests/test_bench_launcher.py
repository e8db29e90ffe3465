"""bench.py's own launcher (`python bench.py --gpus N` with no WORLD_SIZE in the environment): host-side behaviour that needs no
GPU.  The N-rank run itself is tests/test_gpu_dp_rehearsal.py::test_bench_launch_forms_rehearsal (one-GPU box, gloo)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['PYTHONPATH'] = ROOT
    return env


def test_the_launcher_path_never_imports_torch():
    """The parent of a self-launched run must not be able to initialise the GPU: importing bench.py pulls in neither torch
    nor the package (children are fresh interpreters, never re-execs of a process that touched HIP)."""
    code = "import sys, bench; assert 'torch' not in sys.modules and 'geniconet_amd' not in sys.modules, sorted(sys.modules)"
    subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=_env(), check=True, timeout=120)


def test_a_failing_rank_fails_the_launcher():
    """No GPU in this container: every rank exits with 'no GPU visible'; the launcher must forward a non-zero code (and not
    print a bench line).  With a GPU present this test has nothing to show and passes trivially on rc == 0."""
    import torch
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
                        '--no-cpu-baseline'], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    if torch.cuda.is_available():
        return
    assert r.returncode != 0
    assert 'no GPU visible' in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]


def test_a_world_size_mismatch_is_refused():
    env = dict(_env(), WORLD_SIZE='3', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE=3' in r.stderr


def test_roofline_traffic_is_only_reported_for_the_sources_it_was_measured_on(tmp_path, monkeypatch):
    """bench.committed_traffic: the committed counter profile's HBM bytes are reported only when it records the sha256 of the kernel
    sources of THIS tree (tools/profile_summary.py writes it); a profile of other / unrecorded sources gives None + the reason."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from geniconet_amd import _lib
    prof = tmp_path / 'profiles'
    prof.mkdir()
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    kernel = 'k_conv_dma_sk<64, 128, false>'
    assert bench.committed_traffic(kernel)[0] is None                                   # no profile at all
    entry = {'icn::' + kernel: {'hbm_bytes_per_launch': 123.0}}
    (prof / 'r07_pmc_per_kernel.json').write_text(json.dumps(entry))                    # hash unrecorded (rounds 1-4)
    t, why = bench.committed_traffic(kernel)
    assert t is None and 'unrecorded' in why
    (prof / 'r08_pmc_per_kernel.json').write_text(json.dumps(dict(entry, _kernel_sources_sha256='0' * 64)))
    t, why = bench.committed_traffic(kernel)
    assert t is None and 're-run tools/profile_round.sh' in why                         # other sources
    (prof / 'r09_pmc_per_kernel.json').write_text(json.dumps(dict(entry, _kernel_sources_sha256=_lib.source_sha256())))
    t, why = bench.committed_traffic(kernel)
    assert t == 123.0 and 'r09_pmc_per_kernel.json' in why
    assert bench.committed_traffic('k_other')[0] is None                                # a kernel the profile does not have
