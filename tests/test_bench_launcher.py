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


def test_a_pricing_build_or_an_overridden_library_cannot_pass_for_the_product(tmp_path, monkeypatch):
    """(1) a library with ICN_EXP bits is refused at load unless ICN_ALLOW_EXP=1; (2) the loaded library reports its path,
    whether ICN_LIB_PATH overrode the in-tree build, and its build switches (bench.py prints them in the line); (3) committed counter
    traffic is not reported for an overridden library."""
    import json
    import pytest
    sys.path.insert(0, ROOT)
    import bench
    from geniconet_amd import _lib
    with pytest.raises(RuntimeError, match='pricing build'):
        _lib.refuse_pricing_build(0x0004 | (4 << 16), '/x/libicn_exp4.so', allow=False)
    _lib.refuse_pricing_build(0x0004 | (4 << 16), '/x/libicn_exp4.so', allow=True)
    _lib.refuse_pricing_build(4 << 16, '/x/libicn.so', allow=False)                       # the product: ICN_EXP = 0
    info = _lib.build_info()
    assert info['exp'] == 0 and info['conv_waves_default'] in (4, 8) and info['path'].endswith('.so')
    assert info['overridden'] == bool(os.environ.get('ICN_LIB_PATH'))
    prof = tmp_path / 'profiles'
    prof.mkdir()
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    kernel = 'k_conv_b3_sk<128, 128, 8>'
    entry = {'icn::' + kernel: {'hbm_bytes_per_launch': 5.0}, '_kernel_sources_sha256': _lib.source_sha256(), '_step_hbm_bytes': 1.5e10,
             '_arith': _lib.get_arith()}
    (prof / 'r09_pmc_per_kernel.json').write_text(json.dumps(entry))
    monkeypatch.setattr(_lib, 'build_info', lambda: dict(info, overridden=False))
    assert bench.committed_traffic(kernel)[0] == 5.0
    assert bench.committed_step_traffic()[0] == 15000000000
    other = 'f32' if _lib.get_arith() == 'bf16x3' else 'bf16x3'
    (prof / 'r09_pmc_per_kernel.json').write_text(json.dumps(dict(entry, _arith=other)))
    t, why = bench.committed_step_traffic()
    assert t is None and other in why                                                   # counters of the other arithmetic
    monkeypatch.setattr(_lib, 'build_info', lambda: dict(info, overridden=True))
    t, why = bench.committed_traffic(kernel)
    assert t is None and 'ICN_LIB_PATH' in why
    assert bench.committed_step_traffic()[0] is None


def test_sigterm_to_the_launcher_stops_its_ranks(tmp_path):
    """a SIGTERM to the launcher alone (a driver's timeout on just that pid) must stop the rank processes it started
    -- exact PIDs, no pattern -- and exit non-zero.  Ranks here are stand-ins that record their pid and sleep (no GPU needed)."""
    import signal
    import time
    child = tmp_path / 'rank.py'
    child.write_text("import os, time\nopen(os.path.join(%r, 'pid%%s' %% os.environ['RANK']), 'w').write(str(os.getpid()))\ntime.sleep(120)\n" % str(tmp_path))
    helper = tmp_path / 'launch.py'
    helper.write_text("import sys, argparse\nsys.path.insert(0, %r)\nimport bench\nbench.__file__ = %r\n"
                      "sys.exit(bench.launch_ranks(argparse.Namespace(gpus=3), []))\n" % (ROOT, str(child)))
    p = subprocess.Popen([sys.executable, str(helper)], cwd=ROOT, env=_env())
    deadline = time.time() + 60
    while time.time() < deadline and len([f for f in os.listdir(tmp_path) if f.startswith('pid')]) < 3:
        time.sleep(0.1)
    pids = [int((tmp_path / ('pid%d' % r)).read_text()) for r in range(3)]
    p.send_signal(signal.SIGTERM)
    rc = p.wait(timeout=30)
    assert rc == 128 + signal.SIGTERM
    for pid in pids:
        for _ in range(50):
            try:
                os.kill(pid, 0)
            except ProcessLookupError:
                break
            time.sleep(0.1)
        else:
            raise AssertionError('rank process %d survived the launcher' % pid)
