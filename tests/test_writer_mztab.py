"""mzTab byte-compatibility, checked COLUMN BY COLUMN: the golden text
(tests/golden/mztab_golden.json) was printed by the reference's own writer
(src/ann_solo/writer.py:40-150, imported in place by tests/golden/make_golden.py) for a set of SSMs
and two configurations. The SSM records of ann_solo_amd (spectrum.SpectrumSpectrumMatch) built from
the same inputs must answer every PSM field with exactly the string in the file, and the package's
Config every ``software[1]-setting`` line (tests/mztab_check.py: parsing and a column -> attribute
table; no writer is restated -- an integration keeps the reference's)."""
import json
import os

import numpy as np
import pytest

import mztab_check as M

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def golden():
    return json.load(open(os.path.join(HERE, 'golden', 'mztab_golden.json')))


def _ssms(golden):
    from ann_solo_amd.spectrum import SpectrumSpectrumMatch
    return [SpectrumSpectrumMatch(**r) for r in golden['ssms']]


@pytest.mark.parametrize('case', ['ann_defaults', 'bf_custom'])
def test_every_field_of_the_reference_file_equals_the_records(golden, case):
    c = golden['cases'][case]
    n = M.check_text_against_records(c['text'], _ssms(golden), c['config'])
    mtd, head, rows = M.parse(c['text'])
    assert n == len(rows) * len(M.PSM_FIELDS) + len(M.settings(mtd))
    assert len(head) == 22 and all(len(r) == 21 for r in rows)          # 23 / 22 cells with the PSH / PSM tag (writer.py:125)
    assert set(M.PSM_FIELDS) <= set(head)
    # mode 'ann' lists the five ANN hyper-parameters after the twenty common settings (writer.py:101-105)
    keys = [k for k, _ in M.settings(mtd)]
    assert ('num_list' in keys) == (c['config']['mode'] == 'ann') and keys[19] == 'mode'


def test_config_dataclass_prints_like_the_reference(golden):
    """The package's own Config (reference option names and defaults) answers every setting line
    of the reference's file."""
    from ann_solo_amd.spectral_library import Config
    c = golden['cases']['ann_defaults']
    cfg = Config.open_search(spectral_library_filename='/data/lib/massivekb.splib',
                 query_filename='/data/run/queries.mgf')
    for k, v in c['config'].items():
        if k != 'out_filename':
            assert cfg[k] == v and str(cfg[k]) == str(v), k
    M.check_text_against_records(c['text'], _ssms(golden), cfg)


def test_ssms_from_batch_answer_the_columns():
    from types import SimpleNamespace
    from ann_solo_amd.spectrum import ssms_from_batch
    res = SimpleNamespace(best_row=np.array([2, -1, 0]), pm_count=np.array([1, 0, 2]),
                          pm_pairs=np.zeros((3, 2, 2), np.uint32))
    res.peak_matches = lambda i: res.pm_pairs[i, :res.pm_count[i]].astype(np.int64)
    qm = [dict(identifier=f'scan={10 - i}', index=i, retention_time=1.5 * i, precursor_charge=2,
               precursor_mz=500.25 + i) for i in range(3)]
    lm = [dict(identifier=100 + r, peptide='PEPTIDEK', precursor_mz=np.float32(400.5 + r),
               is_decoy=r == 2) for r in range(3)]
    ssms = ssms_from_batch(res, qm, lm, scores=[0.9, 0.0, 0.5], q_values=[0.001, 1.0, 0.002])
    assert [s.query_identifier for s in ssms] == ['scan=10', 'scan=8']
    assert ssms[0].is_decoy and ssms[0].library_identifier == 102
    f = M.record_fields(ssms[0])
    assert f['PSM_ID'] == 'scan=10' and f['opt_ms_run[1]_cv_MS:1002217_decoy_peptide'] == '1'
    assert f['spectra_ref'] == 'ms_run[1]:index=0' and f['search_engine_score[1]'] == '0.9'
    assert f['calc_mass_to_charge'] == str(np.float32(402.5)) and f['retention_time'] == '0.0'
    assert sorted(['scan=10', 'scan=8'], key=M.natural_key) == ['scan=8', 'scan=10']       # natural, not lexicographic
