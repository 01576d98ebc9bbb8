"""mzTab byte-compatibility CHECKER (test infrastructure, not product): a restatement of the
reference writer's output format (/root/reference/src/ann_solo/writer.py:40-150) used to prove
that the SSM records of the device path carry everything the reference's own writer prints
(SURVEY.md 8f row 4). An integration keeps the reference's ``writer.write_mztab``
(INTEGRATION.md 4a). Note the reference's header has 23 columns and its rows 22
(``opt_ms_run[1]_num_candidates`` has no value, writer.py:125 vs :129-148) -- reproduced as is.
"""
import logging
import os
import pathlib
import re
from typing import List

REFERENCE_VERSION = '0.3.3'          # /root/reference/src/ann_solo/__init__.py:1

_NSRE = re.compile('([0-9]+)')


def natural_sort_key(s: str):
    """writer.py:16-37."""
    return [int(text) if text.isdigit() else text.lower() for text in re.split(_NSRE, s)]


CONFIG_KEYS = [          # writer.py:93-100
    'resolution', 'min_mz', 'max_mz', 'remove_precursor', 'remove_precursor_tolerance',
    'min_intensity', 'min_peaks', 'min_mz_range', 'max_peaks_used', 'max_peaks_used_library',
    'scaling', 'precursor_tolerance_mass', 'precursor_tolerance_mode',
    'precursor_tolerance_mass_open', 'precursor_tolerance_mode_open', 'fragment_mz_tolerance',
    'allow_peak_shifts', 'fdr', 'fdr_min_group_size', 'mode']
CONFIG_KEYS_ANN = ['bin_size', 'hash_len', 'num_candidates', 'num_list', 'num_probe']


def write_mztab(identifications: List, filename: str, config,
                database_version: str = 'null', version: str = REFERENCE_VERSION) -> str:
    """writer.py:40-150. ``config`` answers ``config[key]`` for the reference's option names
    and has ``query_filename`` / ``spectral_library_filename``; ``database_version`` is
    ``SpectralLibraryReader.get_version()`` (reader.py:289-298: 'null')."""
    if os.path.splitext(filename)[1].lower() != '.mztab':
        filename += '.mztab'
    logging.info('Save identifications to file %s', filename)
    metadata = [
        ('mzTab-version', '1.0.0'),
        ('mzTab-mode', 'Summary'),
        ('mzTab-type', 'Identification'),
        ('mzTab-ID', f'ANN-SoLo_{filename}'),
        ('title', f'ANN-SoLo identification file "{filename}"'),
        ('description', f'Identification results of file '
                        f'"{os.path.split(config["query_filename"])[1]}" against '
                        f'spectral library file '
                        f'"{os.path.split(config["spectral_library_filename"])[1]}"'),
        ('software[1]', f'[MS, MS:1001456, ANN-SoLo, {version}]'),
        ('psm_search_engine_score[1]', '[MS, MS:1001143, search engine specific score for PSMs,]'),
        ('psm_search_engine_score[2]', '[MS, MS:1002354, PSM-level q-value,]'),
        ('ms_run[1]-format', '[MS, MS:1001062, Mascot MGF file,]'),
        ('ms_run[1]-location',
         pathlib.Path(os.path.abspath(config['query_filename'])).as_uri()),
        ('ms_run[1]-id_format', '[MS, MS:1000774, multiple peak list nativeID format,]'),
        ('fixed_mod[1]', '[MS, MS:1002453, No fixed modifications searched,]'),
        ('variable_mod[1]', '[MS, MS:1002454, No variable modifications searched,]'),
        ('false_discovery_rate', f'[MS, MS:1002350, PSM-level global FDR, {config["fdr"]}]'),
    ]
    keys = list(CONFIG_KEYS)
    if config['mode'] == 'ann':
        keys.extend(CONFIG_KEYS_ANN)
    for i, key in enumerate(keys):
        metadata.append((f'software[1]-setting[{i}]', f'{key} = {config[key]}'))
    library_uri = pathlib.Path(os.path.abspath(config['spectral_library_filename'])).as_uri()
    with open(filename, 'w') as f_out:
        for m in metadata:
            f_out.write('\t'.join(['MTD'] + list(m)) + '\n')
        f_out.write('\t'.join([
            'PSH', 'sequence', 'PSM_ID', 'accession', 'unique', 'database', 'database_version',
            'search_engine', 'search_engine_score[1]', 'search_engine_score[2]', 'modifications',
            'retention_time', 'charge', 'exp_mass_to_charge', 'calc_mass_to_charge',
            'spectra_ref', 'pre', 'post', 'start', 'end',
            'opt_ms_run[1]_cv_MS:1003062_spectrum_index',
            'opt_ms_run[1]_cv_MS:1002217_decoy_peptide', 'opt_ms_run[1]_num_candidates']) + '\n')
        for ssm in sorted(identifications, key=lambda s: natural_sort_key(s.query_identifier)):
            f_out.write('\t'.join([
                'PSM', ssm.sequence, str(ssm.query_identifier), 'null', 'null', library_uri,
                database_version, '[MS, MS:1001456, ANN SoLo,]', str(ssm.search_engine_score),
                str(ssm.q), 'null', str(ssm.retention_time), str(ssm.charge),
                str(ssm.exp_mass_to_charge), str(ssm.calc_mass_to_charge),
                f'ms_run[1]:index={ssm.query_index}', 'null', 'null', 'null', 'null',
                str(ssm.library_identifier), f'{ssm.is_decoy:d}']) + '\n')
    return filename
