"""A stand-in with the BEHAVIOUR of the reference's configuration singleton
(/root/reference/src/ann_solo/config.py:17-294) for tests: an argparse parser holding the
reference's option names, types and defaults (as data below), ``parse(args)`` filling
``_namespace``, ``__getattr__`` answering from it -- ``RuntimeError`` before ``parse``,
``KeyError`` for an option the parser does not define (config.py:285-291) -- and
``__getitem__``. ``additive=True`` applies the patch INTEGRATION.md 4a gives for the reference's
``Config.__init__``: one call of ``ann_solo_amd.config.add_arguments``."""
import argparse

# (flag, default, type / 'flag', required) -- config.py:52-267, the options of the search path
_OPTIONS = [
    ('--resolution', None, int), ('--min_mz', 11, int), ('--max_mz', 2010, int),
    ('--remove_precursor', False, 'flag'), ('--remove_precursor_tolerance', 0, float),
    ('--min_intensity', 0.01, float), ('--min_peaks', 10, int), ('--min_mz_range', 250, float),
    ('--max_peaks_used', 50, int), ('--max_peaks_used_library', 50, int),
    ('--scaling', 'rank', str), ('--precursor_tolerance_mass', None, float),
    ('--precursor_tolerance_mode', None, str), ('--precursor_tolerance_mass_open', None, float),
    ('--precursor_tolerance_mode_open', None, str), ('--fragment_mz_tolerance', None, float),
    ('--allow_peak_shifts', False, 'flag'), ('--fdr', 0.01, float), ('--model', 'rf', str),
    ('--fdr_min_group_size', 100, int), ('--mode', 'ann', str), ('--bin_size', 0.04, float),
    ('--hash_len', 800, int), ('--num_candidates', 1024, int), ('--batch_size', 16384, int),
    ('--num_list', 256, int), ('--num_probe', 128, int), ('--no_gpu', False, 'flag'),
    ('--add_decoys', False, 'flag'), ('--fragment_tol_mode', 'ppm', str),
]
_REQUIRED = {'--precursor_tolerance_mass', '--precursor_tolerance_mode', '--fragment_mz_tolerance'}


class RefConfig:
    def __init__(self, additive: bool = False):
        self._parser = argparse.ArgumentParser()
        for pos in ('spectral_library_filename', 'query_filename', 'out_filename'):
            self._parser.add_argument(pos)
        for flag, default, typ in _OPTIONS:
            if typ == 'flag':
                self._parser.add_argument(flag, action='store_true')
            else:
                self._parser.add_argument(flag, default=default, type=typ, required=flag in _REQUIRED)
        if additive:
            from ann_solo_amd.config import add_arguments
            add_arguments(self._parser)
        self._namespace = None

    def parse(self, args_str=None) -> None:
        if isinstance(args_str, str):
            args_str = args_str.split()
        self._namespace = vars(self._parser.parse_args(args_str))

    def __getattr__(self, option):
        if option.startswith('__') or option in ('_parser', '_namespace'):
            raise AttributeError(option)
        if self._namespace is None:
            raise RuntimeError('The configuration has not been initialized')
        return self._namespace[option]

    def __getitem__(self, item):
        return self.__getattr__(item)
