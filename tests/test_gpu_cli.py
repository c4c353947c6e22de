"""Steps 12 -> 13 -> 14 end to end through the command-line drivers on a small .npz cube (GPU)."""
import os

import numpy as np
import pytest
import yaml

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _time_cube(nt, nil, nxl, missing, seed=3):
    rng = np.random.default_rng(seed)
    t = np.arange(nt)[:, None, None]
    il = np.arange(nil)[None, :, None] / nil
    xl = np.arange(nxl)[None, None, :] / nxl
    x = np.zeros((nt, nil, nxl))
    for _ in range(4):   # dipping events: band-limited in time, plane waves in space
        f, k1, k2, a = rng.uniform(0.05, 0.2), rng.integers(-3, 4), rng.integers(-3, 4), rng.standard_normal()
        x += a * np.cos(2 * np.pi * (f * t + k1 * il + k2 * xl))
    fold = (rng.random((nil, nxl)) >= missing).astype(np.uint8) * rng.integers(1, 4, (nil, nxl)).astype(np.uint8)
    return (x * (fold > 0)).astype(np.float32), fold


def test_fft_pocs_ifft_pipeline(tmp_path, monkeypatch):
    from oracle import pocs_oracle as orc
    from pseudo_3d_interpolation_amd import cube_POCS_interpolation_3D as step13
    from pseudo_3d_interpolation_amd import cube_apply_FFT as step12
    from pseudo_3d_interpolation_amd import cube_apply_IFFT as step14
    from pseudo_3d_interpolation_amd.cube_io import Cube, open_cube, save_cube
    from pseudo_3d_interpolation_amd.functions.POCS import release_plans

    nt, nil, nxl, dt, t0 = 48, 32, 64, 0.05, 7.0
    x, fold = _time_cube(nt, nil, nxl, 0.5)
    cube = Cube({'env': x, 'fold': fold}, {'env': ('twt', 'iline', 'xline'), 'fold': ('iline', 'xline')},
                {'twt': t0 + dt * np.arange(nt), 'iline': np.arange(nil), 'xline': np.arange(nxl)},
                {'long_name': 'test cube', 'description': 'synthetic', 'history': 'made;', 'text': ''}, {},
                {'twt': {'units': 'ms'}})
    path = save_cube(cube, str(tmp_path / 'cube_twt.npz'))
    nc_yml = tmp_path / 'netcdf.yml'
    nc_yml.write_text(yaml.safe_dump({'attrs_freq': {'data': {'units': 'amplitude'}, 'new_dim': {'units': 'kHz'}},
                                      'attrs_time': {'env': {'units': 'amplitude'}, 'twt': {'units': 'ms', 'spacing': dt}}}))
    pocs_yml = tmp_path / 'pocs.yml'
    metadata = dict(transform_kind='fft', niter=15, eps=0, thresh_op='soft', thresh_model='exponential', decay_kind='values',
                    p_max=0.99, p_min=0.1, alpha=1.0, sqrt_decay=False, version='regular', verbose=False)
    pocs_yml.write_text(yaml.safe_dump({'dim': 'freq_twt', 'var': 'freq_env', 'batch_chunk': 10, 'n_workers': 4, 'processes': True,
                                        'threads_per_worker': 1, 'memory_limit': '2GB', 'output_runtime_results': True,
                                        'metadata': metadata}))

    # step 12
    step12.main(['12_cube_apply_FFT', path, '--params_netcdf', str(nc_yml), '--compute_real'])
    fpath = str(tmp_path / 'cube_freq.npz')
    fcube = open_cube(fpath)
    f = np.fft.rfftfreq(nt, dt)
    X = np.fft.rfft(x.astype(np.float64), axis=0) * (dt * np.exp(-2j * np.pi * f * t0))[:, None, None]
    assert fcube.dims['freq_env'] == ('freq_twt', 'iline', 'xline') and fcube.data_vars['freq_env'].dtype == np.complex64
    assert np.allclose(fcube.coords['freq_twt'], f)
    assert rel_l2(fcube.data_vars['freq_env'], X) < 2e-6
    assert 'FFT(env)' in fcube.attrs['history'] and fcube.var_attrs['freq_env']['original_var'] == 'env'

    # step 13
    step13.main(['13_cube_interpolate_POCS', fpath, '--path_pocs_parameter', str(pocs_yml)])
    prefix = 'cube_freq_FFT_soft_niter-15'
    out_dir = tmp_path / prefix
    assert (out_dir / f'parameter_{prefix}.yml').exists() and (out_dir / f'runtimes_{prefix}.txt').exists()
    batch_files = sorted(p for p in os.listdir(out_dir) if p.endswith('.npz'))
    assert len(batch_files) == 3                                      # 25 frequency slices in batches of 10
    lines = (out_dir / f'runtimes_{prefix}.txt').read_text().strip().splitlines()
    assert len(lines) == 25 and all(len(l.split(';')) == 2 + 15 for l in lines if not l.startswith('0;'))
    icube = open_cube(str(tmp_path / f'{prefix}.npz'))
    assert set(icube.data_vars) == {'freq_env_interp.real', 'freq_env_interp.imag', 'fold'}
    Y = icube.data_vars['freq_env_interp.real'] + 1j * icube.data_vars['freq_env_interp.imag']
    mask = np.where(fold <= 1, fold, 1)
    params = {k: v for k, v in metadata.items() if k not in ('verbose',)}
    params['transform_kind'] = 'FFT'
    want = orc.pocs_cube(fcube.data_vars['freq_env'].astype(np.complex128), mask, **params)
    for s in range(want.shape[0]):
        if 0 < s < want.shape[0] - 1:
            assert rel_l2(Y[s], want[s]) < 1e-5, s
        else:
            # DC and Nyquist slices of a real cube are real up to rounding; the imaginary part of the (complex,
            # lexicographic-max) threshold is then rounding noise of either sign, and so is Im of the result
            assert rel_l2(Y[s].real, want[s].real) < 1e-4, s
    assert 'interp_params_keys' in icube.attrs and 'niter' in icube.attrs['interp_params_keys']
    # `batch_chunk` is the unit of the output files; the GPU got all three batches in one call.  One call per batch (P3D_CLI_GROUP_GIB=0): the same bits
    first_batch = open_cube(str(out_dir / batch_files[1]))
    monkeypatch.setenv('P3D_CLI_GROUP_GIB', '0')
    step13.main(['13_cube_interpolate_POCS', fpath, '--path_pocs_parameter', str(pocs_yml)])
    monkeypatch.delenv('P3D_CLI_GROUP_GIB')
    again = open_cube(str(tmp_path / f'{prefix}.npz'))
    assert np.array_equal(again.data_vars['freq_env_interp.real'], icube.data_vars['freq_env_interp.real'])
    assert np.array_equal(again.data_vars['freq_env_interp.imag'], icube.data_vars['freq_env_interp.imag'])
    second = open_cube(str(out_dir / batch_files[1]))
    assert np.array_equal(second.data_vars['freq_env_interp.real'], first_batch.data_vars['freq_env_interp.real'])
    assert np.array_equal(second.coords['freq_twt'], first_batch.coords['freq_twt'])

    # step 14
    step14.main(['14_cube_apply_IFFT', str(tmp_path / f'{prefix}.npz'), '--params_netcdf', str(nc_yml), '--compute_real'])
    tcube = open_cube(str(tmp_path / f'{prefix.replace("freq", "twt")}_interp-freq.npz'))
    y = tcube.data_vars['env']
    assert y.dtype == np.float32 and tcube.dims['env'] == ('twt', 'iline', 'xline')
    want_t = np.fft.irfft(want * (np.exp(2j * np.pi * f * t0) / dt)[:, None, None], n=nt, axis=0)
    assert rel_l2(y, want_t) < 2e-5
    assert np.allclose(tcube.coords['twt'], t0 + dt * np.arange(nt), atol=1e-4)
    # the interpolation filled the empty traces and kept the observed ones
    obs = fold > 0
    assert np.abs(y[:, ~obs]).max() > 0.05 * np.abs(x).max()
    assert rel_l2(y[:, obs], x[:, obs]) < 1e-4
    release_plans()


def test_netcdf_files_through_the_three_steps(tmp_path):
    """SURVEY section 8f row N2 end to end: the same cube as netCDF-4 files -- written, read and merged by cube_io.py's h5py layer
    under the image's interpreter that has h5py -- through steps 12, 13 and 14 on the GPU.  Every array of every product must
    equal, bit for bit, what the .npz run of the same commands in this interpreter produces, and the per-batch files, the merged
    cube, the parameter and the runtime files must all be there under their netCDF names."""
    import subprocess
    import sys
    from test_cube_io_netcdf import CONDA, conda_has_h5py
    if not conda_has_h5py():
        pytest.skip("no interpreter with h5py in this image")
    from pseudo_3d_interpolation_amd import cube_POCS_interpolation_3D as step13
    from pseudo_3d_interpolation_amd import cube_apply_FFT as step12
    from pseudo_3d_interpolation_amd import cube_apply_IFFT as step14
    from pseudo_3d_interpolation_amd.cube_io import Cube, open_cube, save_cube
    from pseudo_3d_interpolation_amd.functions.POCS import release_plans

    nt, nil, nxl, dt, t0 = 24, 32, 64, 0.05, 7.0
    x, fold = _time_cube(nt, nil, nxl, 0.5)
    cube = Cube({'env': x, 'fold': fold}, {'env': ('twt', 'iline', 'xline'), 'fold': ('iline', 'xline')},
                {'twt': t0 + dt * np.arange(nt), 'iline': np.arange(nil), 'xline': np.arange(nxl)},
                {'long_name': 'test cube', 'description': 'synthetic', 'history': 'made;', 'text': ''}, {}, {'twt': {'units': 'ms'}})
    path = save_cube(cube, str(tmp_path / 'cube_twt.npz'))
    (tmp_path / 'netcdf.yml').write_text(yaml.safe_dump({'attrs_freq': {'data': {'units': 'amplitude'}, 'new_dim': {'units': 'kHz'}},
                                                         'attrs_time': {'env': {'units': 'amplitude'}, 'twt': {'units': 'ms', 'spacing': dt}}}))
    metadata = dict(transform_kind='fft', niter=8, eps=0, thresh_op='soft', thresh_model='exponential', decay_kind='values',
                    p_max=0.99, p_min=0.1, alpha=1.0, sqrt_decay=False, version='regular', verbose=False)
    (tmp_path / 'pocs.yml').write_text(yaml.safe_dump({'dim': 'freq_twt', 'var': 'freq_env', 'batch_chunk': 5, 'n_workers': 4, 'processes': True,
                                                       'threads_per_worker': 1, 'memory_limit': '2GB', 'output_runtime_results': True,
                                                       'metadata': metadata}))
    prefix = 'cube_freq_FFT_soft_niter-8'
    # the .npz run, here
    step12.main(['12_cube_apply_FFT', path, '--params_netcdf', str(tmp_path / 'netcdf.yml'), '--compute_real'])
    step13.main(['13_cube_interpolate_POCS', str(tmp_path / 'cube_freq.npz'), '--path_pocs_parameter', str(tmp_path / 'pocs.yml')])
    step14.main(['14_cube_apply_IFFT', str(tmp_path / f'{prefix}.npz'), '--params_netcdf', str(tmp_path / 'netcdf.yml'), '--compute_real'])
    release_plans()
    # the netCDF run, in the interpreter that has h5py
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([CONDA, os.path.join(root, 'tests', 'helpers', 'nc_pipeline.py'), str(tmp_path), prefix], capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0 and 'NC PIPELINE OK' in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
    nc = tmp_path / 'nc'
    batches = sorted(p for p in os.listdir(nc / prefix) if p.endswith('.nc'))
    assert len(batches) == 3 and (nc / prefix / f'parameter_{prefix}.yml').exists() and (nc / prefix / f'runtimes_{prefix}.txt').exists()
    for name in ('cube_freq', prefix, f'{prefix.replace("freq", "twt")}_interp-freq'):
        a, b = open_cube(str(tmp_path / f'{name}.npz')), open_cube(str(nc / f'{name}.back.npz'))
        assert set(a.data_vars) == set(b.data_vars) and a.dims == b.dims, name
        for k in a.data_vars:
            assert a.data_vars[k].dtype == b.data_vars[k].dtype and np.array_equal(a.data_vars[k], b.data_vars[k]), (name, k)
        for k in a.coords:
            assert np.array_equal(a.coords[k], b.coords[k]), (name, k)
        for k in ('history', 'description'):
            assert str(a.attrs.get(k)) == str(b.attrs.get(k)), (name, k)


def test_wavelet_step13_time_domain(tmp_path):
    """transform_kind: wavelet on a time-domain cube (BASELINE configs[3] in miniature): file naming with the
    `_{wavelet}-smooth` suffix (cube_POCS_interpolation_3D.py:266) and results against the wavelet oracle."""
    from oracle import wavelet_oracle as wo
    from pseudo_3d_interpolation_amd import cube_POCS_interpolation_3D as step13
    from pseudo_3d_interpolation_amd.cube_io import Cube, open_cube, save_cube

    nt, nil, nxl, dt = 12, 48, 64, 0.05
    x, fold = _time_cube(nt, nil, nxl, 0.4, seed=5)
    cube = Cube({'env': x, 'fold': fold}, {'env': ('twt', 'iline', 'xline'), 'fold': ('iline', 'xline')},
                {'twt': dt * np.arange(nt), 'iline': np.arange(nil), 'xline': np.arange(nxl)},
                {'long_name': 'test cube', 'description': 'synthetic', 'history': 'made;', 'text': ''}, {}, {'twt': {'units': 'ms'}})
    path = save_cube(cube, str(tmp_path / 'cube_twt.npz'))
    metadata = dict(transform_kind='wavelet', wavelet='db4', niter=8, eps=0, thresh_op='soft', thresh_model='linear',
                    decay_kind='values', p_max=0.9, p_min=0.05, alpha=1.0, sqrt_decay=False, version='regular', verbose=False)
    pocs_yml = tmp_path / 'pocs.yml'
    pocs_yml.write_text(yaml.safe_dump({'dim': 'twt', 'var': 'env', 'batch_chunk': 5, 'n_workers': 1, 'processes': True,
                                        'threads_per_worker': 1, 'memory_limit': '2GB', 'output_runtime_results': False,
                                        'metadata': metadata}))
    step13.main(['13_cube_interpolate_POCS', path, '--path_pocs_parameter', str(pocs_yml)])
    prefix = 'cube_twt_WAVELET_soft_niter-8'
    out_dir = tmp_path / prefix                                         # directory: name before the suffix is appended (:223-228)
    files = sorted(p for p in os.listdir(out_dir) if p.endswith('.npz'))
    assert len(files) == 3 and all(p.startswith(prefix + '_db4-smooth_') for p in files)
    icube = open_cube(str(tmp_path / f'{prefix}.npz'))
    Y = icube.data_vars['env_interp']
    assert Y.dtype == np.float32 and 'WAVELET (time domain)' in icube.attrs['history']
    mask = np.where(fold <= 1, fold, 1)
    params = {k: v for k, v in metadata.items() if k not in ('verbose', 'transform_kind')}
    want = wo.pocs_cube_wavelet(x.astype(np.float64), mask, **params)
    for s in range(nt):
        assert rel_l2(Y[s], want[s]) < 1e-5, s


def test_shearlet_step13_time_domain(tmp_path):
    """transform_kind: shearlet through the step-13 driver (spectra built by functions/shearlets.py, the stand-in for
    FFST.scalesShearsAndSpectra) against the shearlet oracle."""
    from oracle import shearlet_oracle as so
    from pseudo_3d_interpolation_amd import cube_POCS_interpolation_3D as step13
    from pseudo_3d_interpolation_amd.cube_io import Cube, open_cube, save_cube

    nt, nil, nxl, dt = 6, 32, 48, 0.05
    x, fold = _time_cube(nt, nil, nxl, 0.4, seed=9)
    cube = Cube({'env': x, 'fold': fold}, {'env': ('twt', 'iline', 'xline'), 'fold': ('iline', 'xline')},
                {'twt': dt * np.arange(nt), 'iline': np.arange(nil), 'xline': np.arange(nxl)},
                {'long_name': 'test cube', 'description': 'synthetic', 'history': 'made;', 'text': ''}, {}, {'twt': {'units': 'ms'}})
    path = save_cube(cube, str(tmp_path / 'cube_twt.npz'))
    metadata = dict(transform_kind='shearlet', niter=6, eps=0, thresh_op='soft', thresh_model='exponential', decay_kind='values',
                    p_max=0.99, p_min=0.01, alpha=1.0, sqrt_decay=False, version='regular', verbose=False)
    pocs_yml = tmp_path / 'pocs.yml'
    pocs_yml.write_text(yaml.safe_dump({'dim': 'twt', 'var': 'env', 'batch_chunk': 4, 'n_workers': 1, 'processes': True,
                                        'threads_per_worker': 1, 'memory_limit': '2GB', 'output_runtime_results': False,
                                        'metadata': metadata}))
    step13.main(['13_cube_interpolate_POCS', path, '--path_pocs_parameter', str(pocs_yml)])
    prefix = 'cube_twt_SHEARLET_soft_niter-6'
    icube = open_cube(str(tmp_path / f'{prefix}.npz'))
    Y = icube.data_vars['env_interp']
    mask = np.where(fold <= 1, fold, 1)
    params = {k: v for k, v in metadata.items() if k not in ('verbose', 'transform_kind')}
    want = so.pocs_cube_shearlet(x.astype(np.float64), mask, so.scales_shears_and_spectra((nil, nxl)), **params)
    for s in range(nt):
        assert rel_l2(Y[s], want[s]) < 1e-5, s


def test_step13_with_the_documented_example_config(tmp_path):
    """The configuration file printed in the reference's docs (docs/3D/3D_cube_interpolation.md:126-173): frequency-domain cube,
    FFT, hard threshold, 'exponential-1', p_min 'adaptive', alpha 0.75, version 'fast', eps written as 1e-16 (a YAML-1.1 string)."""
    from oracle import pocs_oracle as orc
    from pseudo_3d_interpolation_amd import cube_POCS_interpolation_3D as step13
    from pseudo_3d_interpolation_amd.cube_io import Cube, open_cube, save_cube

    nf, nil, nxl = 9, 64, 96
    mask = orc.synthetic_mask(nil, nxl, 0.6)
    data = (np.stack([orc.synthetic_slice(nil, nxl, 30 + s) for s in range(nf)]) * mask).astype(np.complex64)
    fold = (mask * 3).astype(np.uint8)
    cube = Cube({'freq_env': data, 'fold': fold}, {'freq_env': ('freq_twt', 'iline', 'xline'), 'fold': ('iline', 'xline')},
                {'freq_twt': np.arange(nf) * 0.1, 'iline': np.arange(nil), 'xline': np.arange(nxl)},
                {'long_name': 'test cube', 'description': 'synthetic', 'history': 'made;', 'text': ''}, {}, {})
    path = save_cube(cube, str(tmp_path / 'cube_freq.npz'))
    (tmp_path / 'pocs.yml').write_text("""
dim: 'freq_twt'
var: 'freq_env'
batch_chunk: 20
n_workers: 12
processes: True
threads_per_worker: 1
memory_limit: '2.5GB'
metadata:
  transform_kind: 'FFT'
  niter: 50
  eps: 1e-16
  thresh_op: 'hard'
  thresh_model: 'exponential-1'
  decay_kind: 'values'
  p_max: 0.99
  p_min: 'adaptive'
  alpha: 0.75
  sqrt_decay: False
  version: 'fast'
  verbose: False
apply_filter: 'gauss'
output_runtime_results: False
""")
    step13.main(['13_cube_interpolate_POCS', path, '--path_pocs_parameter', str(tmp_path / 'pocs.yml')])
    icube = open_cube(str(tmp_path / 'cube_freq_FFT_hard_niter-50.npz'))
    Y = icube.data_vars['freq_env_interp.real'] + 1j * icube.data_vars['freq_env_interp.imag']
    params = dict(niter=50, eps=1e-16, thresh_op='hard', thresh_model='exponential-1', decay_kind='values', p_max=0.99, p_min='adaptive',
                  alpha=0.75, sqrt_decay=False, version='fast')
    infos = []
    want = orc.pocs_cube(data.astype(np.complex128), mask, infos=infos, **params)
    errs = [rel_l2(Y[s], want[s]) for s in range(nf)]
    # hard threshold: a coefficient within float32 round-off of tau can fall on the other side; such a flip
    # costs ~1e-4 on a 64 x 96 slice.  Most slices see none.
    assert max(errs) <= 2e-4, errs
    assert sorted(errs)[nf // 2] <= 1e-5, errs


def test_device_resident_pipeline_matches_the_three_steps():
    """pipeline.interpolate_time_cube (time -> frequency -> POCS -> time on device buffers) = the three host-level steps."""
    from pseudo_3d_interpolation_amd import _ffi, pipeline
    from pseudo_3d_interpolation_amd.functions import POCS as P
    nt, nil, nxl, dt, t0 = 40, 64, 48, 0.004, 0.1
    x, fold = _time_cube(nt, nil, nxl, 0.5, seed=11)
    mask = np.where(fold <= 1, fold, 1)
    kw = dict(niter=12, thresh_op='soft', thresh_model='exponential', eps=1e-9, alpha=0.9, p_max=0.99, p_min=1e-2)
    for real_only, nfft in ((True, None), (False, None), (True, 48)):
        F = _ffi.time2freq(x, dt, t0, nfft=nfft, real_only=real_only)
        G = P.pocs_cube(F, mask, **kw)
        want = _ffi.freq2time(G, dt, t0, nfft=nfft or nt, real_only=real_only)
        res = []
        got = pipeline.interpolate_time_cube(x, mask, dt, t0, nfft=nfft, real_only=real_only, results=res, batch_slices=7, **kw)
        assert got.shape == want.shape and got.dtype == np.float32
        assert rel_l2(got, want) <= 1e-6
        assert len(res) == F.shape[0]
