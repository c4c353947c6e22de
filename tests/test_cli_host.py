"""CPU-only checks of the step 12-14 drivers: command lines, filter windows (hand-derived known answers -- the
reference has no tests and its drivers need xarray/xrft, which are not installable here), file naming, cube I/O."""
import os

import numpy as np
import pytest

from pseudo_3d_interpolation_amd import cube_POCS_interpolation_3D as step13
from pseudo_3d_interpolation_amd import cube_apply_FFT as step12
from pseudo_3d_interpolation_amd import cube_apply_IFFT as step14
from pseudo_3d_interpolation_amd.cube_io import Cube, open_cube, save_cube
from pseudo_3d_interpolation_amd.functions import utils


def test_command_lines_match_the_reference():
    a = step13.define_input_args().parse_args(['cube.nc', '--path_pocs_parameter', 'p.yml', '--path_output_dir', 'out', '-V', '2'])
    assert (a.path_cube, a.path_pocs_parameter, a.path_output_dir, a.verbose) == ('cube.nc', 'p.yml', 'out', 2)
    with pytest.raises(SystemExit):
        step13.define_input_args().parse_args(['cube.nc'])          # --path_pocs_parameter is required
    b = step12.define_input_args().parse_args(['c.nc', '--params_netcdf', 'n.yml', '--compute_real', '--upsampling-factor', '2',
                                                '--filter', 'lowpass', '--filter_freqs', '1200', '1500', '--drop-filtered-freq', '-V'])
    assert b.prefix == 'freq' and b.compute_real and b.upsampling_factor == 2 and b.filter == 'lowpass'
    assert b.filter_freqs == [1200, 1500] and b.drop_filtered_freq and b.verbose == 1
    with pytest.raises(SystemExit):
        step12.define_input_args().parse_args(['c.nc', '--params_netcdf', 'n.yml', '--filter', 'notch'])
    c = step14.define_input_args().parse_args(['c.nc', '--params_netcdf', 'n.yml', '--rescale-envelope'])
    assert c.rescale_envelope and not c.compute_real and c.verbose == 0


def test_stopband_known_answers():
    # 3 samples in the transition band -> Hann window of 7 points: 0, .25, .75, 1, .75, .25, 0
    assert np.allclose(step12._get_stopband(3, 'highpass'), [0.25, 0.75, 1.0])
    assert np.allclose(step12._get_stopband(3, 'lowpass'), [1.0, 0.75, 0.25])
    assert step12._get_stopband(0, 'lowpass').size == 0
    assert step12._get_const_values('highpass') == (0, 1)
    assert step12._get_const_values('lowpass') == (1, 0)
    assert step12._get_const_values('bandpass') == (0, 0)


def test_filter_windows():
    f = np.arange(10.0)
    low = step12.get_freq_filter_win([2, 4], f, filter_type='lowpass')          # 3 samples in [2, 4]
    assert np.allclose(low, [1, 1, 1.0, 0.75, 0.25, 0, 0, 0, 0, 0])
    high = step12.get_freq_filter_win([6, 8], f, filter_type='highpass')
    assert np.allclose(high, [0, 0, 0, 0, 0, 0, 0.25, 0.75, 1.0, 1])
    band = step12.get_freq_filter_win([1, 3, 6, 8], f, filter_type='bandpass')
    assert np.allclose(band, [0, 0.25, 0.75, 1, 1, 1, 1, 0.75, 0.25, 0])
    assert band.shape == f.shape and band.min() >= 0 and band.max() <= 1
    assert np.array_equal(step12.get_freq_filter_mask(f, freqs=[2, 4], filter_type='lowpass'), f <= 4)
    assert np.array_equal(step12.get_freq_filter_mask(f, freqs=[6, 8], filter_type='highpass'), f >= 6)
    assert np.array_equal(step12.get_freq_filter_mask(f, freqs=[1, 3, 6, 8], filter_type='bandpass'), (f >= 1) & (f <= 8))
    with pytest.raises(AssertionError):
        step12.get_freq_filter_mask(f, freqs=[1, 2, 3], filter_type='lowpass')


def test_file_naming_and_runtime_files(tmp_path):
    p = step13.create_file_path(np.array([0.5, 0.75, 1.0]), prefix='cube_FFT_hard_niter-50', root_path=str(tmp_path))
    assert os.path.basename(p) == 'cube_FFT_hard_niter-50_00.500_01.000.nc'
    p = step13.create_file_path(np.float64(12.25), prefix='x', root_path='.', suffix='.npz')
    assert os.path.basename(p) == 'x_12.250_12.250.npz'
    (tmp_path / 'slice-0000-0002.out').write_text('3;0.1;1e-3;1e-4;1e-5\n')
    (tmp_path / 'slice-0002-0004.out').write_text('2;0.2;1e-3;1e-4\n')
    step13.combine_runtime_results(str(tmp_path), prefix='run')
    lines = sorted((tmp_path / 'runtimes_run.txt').read_text().splitlines())
    assert lines == ['2;0.2;1e-3;1e-4', '3;0.1;1e-3;1e-4;1e-5']


def test_split_complex_and_cube_roundtrip(tmp_path):
    z = (np.arange(24).reshape(2, 3, 4) * (1 + 2j)).astype(np.complex64)
    c = Cube({'freq_env_interp': z, 'fold': np.ones((3, 4), np.uint8)},
             {'freq_env_interp': ('freq_twt', 'iline', 'xline'), 'fold': ('iline', 'xline')},
             {'freq_twt': np.array([0.0, 0.5]), 'iline': np.arange(3), 'xline': np.arange(4)},
             {'history': 'a;'}, {'freq_env_interp': {'units': 'amplitude'}}, {'freq_twt': {'units': 'kHz', 'nfft': 4}})
    assert c.slice_dim() == 'freq_twt'
    step13.split_complex_variable(c, 'freq_env_interp')
    assert set(c.data_vars) == {'freq_env_interp.real', 'freq_env_interp.imag', 'fold'}
    assert np.array_equal(c.data_vars['freq_env_interp.imag'], z.imag) and c.data_vars['freq_env_interp.real'].dtype == np.float32
    path = save_cube(c, str(tmp_path / 'cube.npz'))
    d = open_cube(path)
    assert d.dims == c.dims and d.attrs == c.attrs and d.coord_attrs['freq_twt']['nfft'] == 4
    for k in c.data_vars:
        assert np.array_equal(d.data_vars[k], c.data_vars[k]) and d.data_vars[k].dtype == c.data_vars[k].dtype
    with pytest.raises(ValueError):
        save_cube(c, str(tmp_path / 'cube.segy'))


def test_utils(capsys):
    utils.xprint('hello', kind='info', verbosity=0)
    assert capsys.readouterr().out == ''
    utils.xprint('hello', kind='info', verbosity=1)
    assert '[INFO]' in capsys.readouterr().out
    utils.xprint('careful', kind='warning', verbosity=0)
    assert '[WARN]' in capsys.readouterr().out
    a = np.array([2.0, 4.0, 6.0])
    assert np.allclose(utils.rescale(a), [0, 0.5, 1])
    assert np.allclose(utils.rescale_dask(a, amin=0.0, amax=8.0), [0.25, 0.5, 0.75])
    assert utils.rescale_dask(np.ones(3)) is not None
