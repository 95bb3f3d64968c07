"""bench.py's launcher contract, checked without a GPU: `--gpus N` with no torch.distributed environment starts N ranks
under torch.distributed.run as child processes BEFORE anything touches the GPU, and a launcher/--gpus mismatch is
refused (`--gpus` was parsed and never used)."""
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    return bench


def test_gpus_n_relaunches_under_torch_distributed_run(monkeypatch):
    bench = _bench()
    calls = []

    def fake_run(cmd, env=None, **kw):
        calls.append((cmd, env))
        return types.SimpleNamespace(returncode=7)
    monkeypatch.setattr(bench.subprocess, 'run', fake_run)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3', '--warmup', '1'])
    import torch
    monkeypatch.setattr(torch.cuda, 'set_device', lambda *a: (_ for _ in ()).throw(AssertionError('GPU touched before the relaunch')))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                   # the children's exit code is handed back
    (cmd, env), = calls
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nproc-per-node=4' in cmd and '--nnodes=1' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert os.path.abspath(os.path.join(ROOT, 'bench.py')) in cmd
    assert cmd[-6:] == ['--gpus', '4', '--steps', '3', '--warmup', '1']
    assert env.get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'


def test_rank_count_mismatch_is_refused(monkeypatch):
    bench = _bench()
    monkeypatch.setenv('WORLD_SIZE', '2')
    monkeypatch.setenv('RANK', '0')
    monkeypatch.setenv('LOCAL_RANK', '0')
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8'])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert 'started 2 ranks' in str(e.value.code)


def test_c4_volume_is_a_function_of_the_coordinates():
    """Every rank of every world size must generate the same volume: a slab equals the same planes of a taller one."""
    import torch
    bench = _bench()
    a = bench.synth_volume_slab(0, 70, 24, 40, 'cpu')
    b = bench.synth_volume_slab(33, 61, 24, 40, 'cpu')
    assert torch.equal(a[33:61], b)
    assert a.min() >= 1 and a.float().std() > 10               # non-zero, textured data
