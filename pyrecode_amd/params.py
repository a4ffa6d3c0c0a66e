"""Run-level (InitParams) and file-level (InputParams) parameters.
Same interface as reference pyrecode/params.py: InitParams :7-190, InputParams :193-570 (`key = int` text files,
config/README.md).  Host side only; nothing here touches the GPU."""
from pathlib import Path  # noqa: F401  (kept for callers that pass Path objects)

from .misc import map_dtype


class InitParams:
    """Validated holder of the initialisation arguments (mode, file names, verbosity, ...)."""

    _FIELDS = ("mode", "verbosity", "validation_frame_gap", "image_filename", "calibration_filename", "params_filename",
               "output_directory", "log_filename", "run_name", "use_c", "directory_path", "max_count", "chunk_time_in_sec")

    def __init__(self, mode, output_directory, image_filename='', directory_path='', calibration_filename='',
                 params_filename='', validation_frame_gap=-1, log_filename='recode.log', run_name='run',
                 verbosity=0, use_c=False, max_count=-1, chunk_time_in_sec=0):
        self._mode = mode.strip().lower()
        self._verbosity = verbosity
        self._validation_frame_gap = validation_frame_gap
        self._image_filename = image_filename
        self._calibration_filename = calibration_filename
        self._params_filename = params_filename
        self._output_directory = output_directory
        self._log_filename = log_filename
        self._run_name = run_name
        self._use_c = use_c
        self._directory_path = directory_path
        self._max_count = max_count
        self._chunk_time_in_sec = chunk_time_in_sec
        if not self._validate_init_params():
            self.show_usage()
            raise ValueError('Invalid initialization parameters')

    def validate(self):
        self._validate_init_params()

    def _validate_init_params(self):
        if self._output_directory == '':
            print('Output Directory cannot be empty')
            return False
        if self._mode not in ('batch', 'stream'):
            print("Unknown mode: mode can only be 'batch' or 'stream'")
            return False
        if self._mode == 'batch' and self._image_filename == '':
            print('Image filename cannot be empty')
            return False
        self._verbosity = min(max(self._verbosity, 0), 2)
        return True

    @staticmethod
    def show_usage():
        print("See documentation at https://github.com/NDLOHGRP/pyReCoDe for usage details")


def _ro(name):
    return property(lambda self: getattr(self, "_" + name))


for _f in InitParams._FIELDS:
    setattr(InitParams, _f, _ro(_f))


class InputParams:
    """`key = int` parameters of one dataset.  Unknown keys are rejected on load; validate() range-checks and derives
    the numpy source / target dtypes."""

    _KEYS = ('reduction_level', 'rc_operation_mode', 'calibration_threshold_epsilon', 'target_bit_depth',
             'source_bit_depth', 'num_cols', 'num_rows', 'num_frames', 'frame_offset', 'num_calibration_frames',
             'calibration_frame_offset', 'keep_part_files', 'num_threads', 'l2_statistics', 'l4_centroiding',
             'compression_scheme', 'compression_level', 'source_file_type', 'source_header_length',
             'keep_calibration_data', 'calibration_file_type', 'source_data_type', 'target_data_type',
             'source_numpy_dtype', 'target_numpy_dtype')

    def __init__(self):
        self._param_map = {k: -1 for k in self._KEYS}

    def load(self, params_filename):
        assert params_filename != '', 'Params filename missing'
        with open(params_filename) as fp:
            for line in fp:
                if line.strip() == '' or line.startswith('#'):
                    continue
                key, _, value = line.partition('=')
                key = key.strip().lower()
                assert key in self._param_map, 'Unknown parameter: ' + key
                self._param_map[key] = int(value.strip().lower())

    def _validate_input_params(self):
        p = self._param_map
        binary_like = p['source_file_type'] in (0, 3)

        def bad(msg):
            print(msg)
            return False

        if p['reduction_level'] not in (1, 2, 3, 4):
            return bad('Reduction level must be 1, 2, 3 or 4')
        if p['rc_operation_mode'] not in (0, 1):
            return bad('RC Operation mode can be 0, 1 or 2')
        if p['calibration_threshold_epsilon'] == '':
            return bad('Calibration threshold (epsilon) cannot be empty')
        for key, label in (('source_bit_depth', 'Source bit depth'), ('num_cols', 'Number of columns'),
                           ('num_rows', 'Number of rows'), ('num_frames', 'Number of frames')):
            if p[key] == -1 and binary_like:
                return bad(label + ' cannot be empty when source filetype is binary/other')
        for key, label in (('frame_offset', 'Frame offset'), ('num_calibration_frames', 'Number of calibration'),
                           ('calibration_frame_offset', 'Calibration frame offset'), ('num_threads', 'Number of threads')):
            if not isinstance(p[key], int):
                return bad(label + ' should be an integer')
        if p['keep_part_files'] not in (0, 1):
            return bad('Keep part files must be 0 or 1')
        if p['l2_statistics'] not in (0, 1, 2):
            return bad('L2 statistics must be 0, 1 or 2')
        if p['l4_centroiding'] not in (0, 1, 2, 3):
            return bad('L4 centroiding must be 0, 1, 2 or 3')
        if p['compression_scheme'] not in range(12):
            return bad('Compression scheme must be 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10 or 11')
        if not 0 <= int(p['compression_level']) <= 22:
            return bad('Compression level can be from 0 - 22')
        if p['keep_calibration_data'] not in (0, 1):
            return bad('Keep dark data cannot be either 0 or 1')
        if p['source_file_type'] not in (0, 1, 2, 3):
            return bad('Source file type must be 0, 1, 2 or 3')
        if binary_like and (p['source_header_length'] == -1 or not isinstance(p['source_header_length'], int)):
            return bad('Source Header Length cannot be empty or non-integer when source filetype is binary/other')
        if p['calibration_file_type'] not in (0, 1, 2, 3):
            return bad('Calibration filetype must be 0, 1, 2 or 3')
        p['frame_offset'] = max(p['frame_offset'], 0)
        p['num_threads'] = max(p['num_threads'], 1)
        if p['source_data_type'] not in (0, 1, 2):
            return bad('Source data type must be 0, 1, or 2')
        if p['target_data_type'] not in (0, 1, 2):
            return bad('Target data type must be 0, 1, or 2')
        if p['target_bit_depth'] == -1:
            p['target_bit_depth'] = p['source_bit_depth']
        p['source_numpy_dtype'] = map_dtype(p['source_data_type'], p['source_bit_depth'])
        p['target_numpy_dtype'] = map_dtype(p['target_data_type'], p['target_bit_depth'])
        return True

    def validate(self):
        return self._validate_input_params()

    def serialize(self, filename):
        with open(filename, 'w') as f:
            for key, value in self._param_map.items():
                f.write(key + ' = ' + str(value) + '\n')


def _rw(key, writable):
    def getter(self):
        return self._param_map[key]

    def setter(self, value):
        self._param_map[key] = value
    return property(getter, setter) if writable else property(getter)


# attribute name -> (param key, has a setter in the reference)
for _attr, (_key, _w) in {
    'reduction_level': ('reduction_level', True), 'rc_operation_mode': ('rc_operation_mode', False),
    'calibration_threshold_epsilon': ('calibration_threshold_epsilon', False),
    'target_bit_depth': ('target_bit_depth', False), 'source_bit_depth': ('source_bit_depth', False),
    'num_cols': ('num_cols', True), 'num_rows': ('num_rows', True), 'num_frames': ('num_frames', True),
    'nx': ('num_cols', True), 'ny': ('num_rows', True), 'nz': ('num_frames', True),
    'frame_offset': ('frame_offset', False), 'num_calibration_frames': ('num_calibration_frames', False),
    'calibration_frame_offset': ('calibration_frame_offset', False), 'keep_part_files': ('keep_part_files', False),
    'num_threads': ('num_threads', False), 'l2_statistics': ('l2_statistics', False),
    'l4_centroiding': ('l4_centroiding', False), 'L2_statistics': ('l2_statistics', False),
    'L4_centroiding': ('l4_centroiding', False), 'compression_scheme': ('compression_scheme', False),
    'compression_level': ('compression_level', False), 'keep_calibration_data': ('keep_calibration_data', False),
    'source_file_type': ('source_file_type', False), 'source_header_length': ('source_header_length', False),
    'calibration_file_type': ('calibration_file_type', False), 'source_data_type': ('source_data_type', True),
    'target_data_type': ('target_data_type', True), 'source_numpy_dtype': ('source_numpy_dtype', False),
    'target_numpy_dtype': ('target_numpy_dtype', False),
}.items():
    setattr(InputParams, _attr, _rw(_key, _w))
