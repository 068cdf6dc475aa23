"""blackbox_amd.qc -- quality-control flags on the reduction keywords.

Host-side mirror of the reference's qc.py: `qc_check` (qc.py:15-516) compares header values with
per-telescope ranges and writes `QC-FLAG`, `DUMCAT` and the `QC{RED,ORA,YEL}n` keywords that the
rest of the BlackBOX / BlackGEM pipeline (database ingestion) keys off; `run_qc_check`
(qc.py:520-551) reduces that to one colour.  blackbox_reduce calls it after every calibration
block (blackbox.py:1095, 1629, 1709, 1777, 2000).

The ranges themselves are deployment settings (the reference keeps them in Settings/set_qc.py,
1264 lines for all telescopes and catalog keywords).  This module looks for that module first
(`import set_qc`, as the reference does) and otherwise uses QC_RANGE below, which restates only the
entries for the keywords the reduction path of this package writes (a2-a13), with the lines they
come from.  Pass `qc_range=` to use any other table.

Not mirrored: the dummy-catalog creation (`cat_dummy`, zogy.format_cat [EXT]); a non-None
`cat_dummy` only sets DUMCAT = True like the reference does before it builds the table.
"""
import logging

import numpy as np

log = logging.getLogger(__name__)

COLORS = ['green', 'yellow', 'orange', 'red']      # qc.py:136
N_STD = [2, 4, 7]                                  # qc.py:146: 'sigma' ranges = E +- n * STD


def _entry(default, val_type, val_range, comment, key_type='full', pos=False):
    return {'default': default, 'val_type': val_type, 'val_range': val_range, 'key_type': key_type,
            'pos': pos, 'comment': comment}


def _reduction_keys(mbias, biasmean, rdnoise):
    """the processing flags and level keywords of blackbox_reduce (set_qc.py:106-122, 136, 177,
    249-253 for ML1; 447-463, 475-479, 529-533 for BG)"""
    t = {
        'XTALK-P': _entry(False, 'bool', [True], 'corrected for crosstalk?'),
        'NONLIN-P': _entry(False, 'bool', [False], 'corrected for non-linearity?'),
        'GAIN-P': _entry(False, 'bool', [True], 'corrected for gain?'),
        'OS-P': _entry(False, 'bool', [True], 'corrected for overscan?'),
        'MBIAS-P': _entry(False, 'bool', [mbias], 'corrected for master bias?'),
        'MFLAT-P': _entry(False, 'bool', [True], 'corrected for master flat?'),
        'COSMIC-P': _entry(False, 'bool', [True], 'corrected for cosmics rays?'),
        'SAT-P': _entry(False, 'skip', [True, False], 'processed for satellite trails?'),
        'BIASMEAN': biasmean,
        'RDNOISE': rdnoise,
        'N-INFNAN': _entry('None', 'min_max', [(0, 0), (1, 10), (11, 1e6)], 'number of pixels with infinite/nan values', pos=True),
        'NCOSMICS': _entry('None', 'min_max', [(3, 50), (2, 100), (0, 500)], '[/s] number of cosmic rays identified', pos=True),
        'NSATS': _entry('None', 'min_max', [(0, 10), (10, 20), (20, 100)], 'number of satellite trails identified', pos=True),
    }
    return t


QC_RANGE = {
    'ML1': _reduction_keys(False,
                           _entry('None', 'sigma', [(6450, 100)], 'average all channel means vertical overscan', pos=True),
                           _entry('None', 'min_max', [(5, 11), (5, 13), (5, 15)], 'average all channel sigmas vertical overscan', pos=True)),
    'BG': _reduction_keys(True,
                          _entry('None', 'skip', [(3200, 100)], '[e-] average all channel means vertical overscan', pos=True),
                          _entry('None', 'min_max', [(5, 14), (5, 17), (5, 20)], '[e-] average all channel sigmas vertical overscan', pos=True)),
}


_DEPLOYED = None


def _table(telescope, qc_range):
    if qc_range is None:
        global _DEPLOYED
        if _DEPLOYED is None:                               # (once: a failing import searches sys.path every time)
            try:
                import set_qc                               # the deployment's full table
                _DEPLOYED = set_qc.qc_range
            except ImportError:
                _DEPLOYED = QC_RANGE
        qc_range = _DEPLOYED
    if telescope in qc_range:
        return qc_range[telescope]
    return qc_range[telescope[0:2]]                         # all BlackGEM telescopes share 'BG' (qc.py:121-125)


def _set(header, key, value, comment, after=None):
    """astropy Header.set(..., after=) or a plain dict"""
    if hasattr(header, 'set') and not isinstance(header, dict):
        header.set(key, value, comment, after=after if (after is not None and after in header) else None)
    else:
        header[key] = (value, comment) if _tuple_style(header) else value


def _tuple_style(header):
    return any(isinstance(v, tuple) for v in header.values())


def _val(header, key):
    v = header[key]
    return v[0] if isinstance(v, tuple) else v


def _check_ranges(value, val_type, val_range, pos):
    """-> (index of the first range that holds the value, or the number of ranges; that number;
    the range text of every step tried) following qc.py:283-352"""
    if val_type == 'sigma':
        val_range = [(val_range[0][0], val_range[0][1] * n) for n in N_STD]
    texts = []
    range_ok = None
    for i, r in enumerate(val_range):
        if val_type in ('exp_abs', 'sigma'):
            ok = np.abs(value - r[0]) <= r[1]
            range_ok = [r[0] - r[1], r[0] + r[1]]
        elif val_type == 'exp_frac':
            ok = np.abs((value - r[0]) / r[0]) <= r[1]
            range_ok = [r[0] * (1. - r[1]), r[0] * (1. + r[1])]
        elif val_type in ('min_max', 'key'):
            ok = (value >= r[0] and value <= r[1])
            range_ok = [r[0], r[1]]
        elif val_type == 'bool':
            ok = (value == r)
            range_ok = r if i == 0 else [range_ok, r]
        else:
            raise ValueError('[val_type] not one of "exp_abs", "exp_frac", "min_max", "bool", "sigma" or "key"')
        if pos and val_type != 'bool':
            range_ok = [max(0, range_ok[0]), max(0, range_ok[1])]
        texts.append('{}'.format(range_ok) if isinstance(range_ok, bool) else '{:g},{:g}'.format(*range_ok))
        if ok:
            return i, len(val_range), texts
    return len(val_range), len(val_range), texts


def qc_check(header, telescope='ML1', keywords=None, check_key_type=None, cat_dummy=None, cat_type=None,
             return_range_comment=False, hide_greens=True, hide_warnings=True, qc_range=None):
    """-> (keywords, colours[, ranges, comments]) of the checked keywords (only the non-green ones
    unless hide_greens=False); writes (T)QC-FLAG, (T)DUMCAT and (T)QC{RED,ORA,YEL}n into [header]"""
    table = _table(telescope, qc_range)
    if keywords is None:
        keywords = list(table.keys())
    filt = _val(header, 'FILTER') if 'FILTER' in header else None
    colors_out, ranges = [], {}
    for key in keywords:
        K = key.upper()
        color = ''
        if K not in table or K not in header:
            if not hide_warnings:
                log.warning('keyword %s not present in %s', key, 'qc_range' if K not in table else 'the input header')
            colors_out.append(color)
            continue
        e = table[K]
        val_type = e['val_type']
        if val_type == 'skip' or (check_key_type is not None and e['key_type'] != check_key_type):
            colors_out.append(color)
            continue
        if K == 'ISTRACKI' and str(_val(header, 'IMAGETYP')).lower() != 'object':
            colors_out.append('green')                       # (qc.py:196: left at its initial colour)
            continue
        val_range = e['val_range']
        if val_type == 'key':
            try:
                val_range = [[eval(v, {'header': header, 'np': np}) if isinstance(v, str) else v for v in item]
                             for item in val_range]
            except Exception:
                log.warning('could not evaluate a range of %s; skipping its quality check', key)
                colors_out.append(color)
                continue
        if isinstance(val_range, dict):
            val_range = val_range[filt]
        value = _val(header, K)
        if value == 'None' or value is None:
            colors_out.append(color)
            continue
        if val_type == 'bool' and isinstance(value, str):
            value = value.strip() == 'T'
        if ('IMAGETYP' in header and 'DEC' in header and str(_val(header, 'IMAGETYP')).lower() == 'object'
                and _val(header, 'DEC') <= -87 and K in ('A-DRA', 'A-DRASTD', 'A-DDEC', 'A-DDESTD')):
            val_range = [tuple(2 * np.array(r)) for r in val_range]       # qc.py:275-277
        idx, nranges, texts = _check_ranges(value, val_type, val_range, e['pos'])
        # inside range i -> colour i; inside none -> red.  The range quoted in the header is the
        # one of the next better colour (qc.py:331-352)
        colors_out.append(COLORS[idx] if idx < nranges else COLORS[-1])
        ranges[K] = texts[0 if idx == 0 else idx - 1]
    colors_arr = np.array(colors_out)
    mask = colors_arr != ''
    if hide_greens:
        mask &= colors_arr != 'green'
    qc_flag = 'green'
    for col in COLORS:
        if col in colors_arr[mask]:
            qc_flag = col
    prefix, label = ('T', 'transient ') if check_key_type == 'trans' else ('', '')
    _set(header, prefix + 'QC-FLAG', qc_flag, '{}QC flag (green|yellow|orange|red)'.format(label),
         after='QC-FLAG' if prefix else None)
    _set(header, prefix + 'DUMCAT', cat_dummy is not None, 'dummy {}catalog without sources?'.format(label),
         after='DUMCAT' if prefix else None)
    if 'QC-FLAG' in header and 'TQC-FLAG' in header:
        main = _val(header, 'QC-FLAG')
        if COLORS.index(qc_flag) < COLORS.index(main):
            _set(header, 'TQC-FLAG', main, 'transient QC flag (green|yellow|orange|red)')
            _set(header, 'TQC{}1'.format(main[0:3].upper()), 'QC-FLAG', 'flag inherited from QC-FLAG', after='TQC-FLAG')
    if cat_dummy is not None:
        # dummy catalogue of type [cat_type] (qc.py:451-503): an empty table whose header carries
        # the image header plus the defaults of the keywords a catalogue of that type would hold
        from .catalogs import format_cat
        header_dummy = dict(header)
        for key, e in table.items():
            if key not in header_dummy and e['key_type'] in (cat_type, 'full'):
                header_dummy[key] = (e['default'], e['comment'])
        format_cat(None, cat_dummy, cat_type=cat_type or 'new', header2add=header_dummy)
    prev = prefix + 'QC-FLAG'
    kw = np.array([k.upper() for k in keywords])
    for col in ('red', 'orange', 'yellow'):
        better = COLORS[COLORS.index(col) - 1]
        for n, k in enumerate(kw[colors_arr == col]):
            name = '{}QC{}{}'.format(prefix, col[0:3].upper(), n + 1)
            _set(header, name, k, '{} range: {}'.format(better, ranges[k]), after=prev)
            prev = name
    keys_out = kw[mask].tolist()
    cols_out = colors_arr[mask].tolist()
    if return_range_comment:
        return keys_out, cols_out, [ranges[k] for k in keys_out], [table[k]['comment'] for k in keys_out]
    return keys_out, cols_out


def run_qc_check(header, telescope, cat_type=None, cat_dummy=None, check_key_type=None, qc_range=None):
    """the most severe colour among the checked keywords (qc.py:520-551)"""
    keys, colors, ranges, comments = qc_check(header, telescope=telescope, cat_type=cat_type, cat_dummy=cat_dummy,
                                              check_key_type=check_key_type, return_range_comment=True,
                                              qc_range=qc_range)
    qc_flag = 'green'
    for col in ('yellow', 'orange', 'red'):
        if col in colors:
            qc_flag = col
    if qc_flag == 'red':
        for k, c, r, cm in zip(keys, colors, ranges, comments):
            if c == 'red':
                log.error('%s flag for keyword: %s, value: %s, allowed range: %s, comment: %s', c, k, _val(header, k), r, cm)
    return qc_flag


# ---- header contract ---------------------------------------------------------------------
def _k(dtype, db, none_ok, htype='full'):
    return {'htype': htype, 'dtype': dtype, 'DB': db, 'None_OK': none_ok}


def _reduction_contract():
    """the entries of verify_header's dictionary (blackbox.py:3004-3058) for the keywords that
    blackbox_reduce writes up to and including the reduced image and its mask; the per-channel
    families are spelled out for all 16 channels (the reference lists channel 1 and 16)"""
    d = {
        'BB-V': _k(str, True, False), 'BB-START': _k(str, True, False), 'KW-V': _k(str, True, False),
        'N-INFNAN': _k(int, True, True),
        'XTALK-P': _k(bool, True, False), 'XTALK-F': _k(str, False, True),
        'NONLIN-P': _k(bool, True, False), 'NONLIN-F': _k(str, False, True),
        'GAIN-P': _k(bool, True, False), 'GAIN': _k(float, False, True),
        'OS-P': _k(bool, True, False), 'BIASMEAN': _k(float, True, True), 'RDNOISE': _k(float, True, True),
        'MBIAS-P': _k(bool, True, False), 'MBIAS-F': _k(str, True, True),
        'SATURATE': _k(float, False, True), 'NOBJ-SAT': _k(int, False, True),
        'MFLAT-P': _k(bool, True, False), 'MFLAT-F': _k(str, True, True),
        'MFRING-P': _k(bool, True, False), 'MFRING-F': _k(str, True, True), 'FRRATIO': _k(float, False, True),
        'COSMIC-P': _k(bool, True, False), 'NCOSMICS': _k(float, True, True),
        'SAT-P': _k(bool, True, False), 'NSATS': _k(int, True, True),
        'REDFILE': _k(str, True, True), 'MASKFILE': _k(str, True, True),
        'DUMCAT': _k(bool, True, False), 'QC-FLAG': _k(str, True, False),
    }
    for c in range(1, 17):
        d['GAIN{}'.format(c)] = _k(float, False, True)
        d['BIASM{}'.format(c)] = _k(float, True, True)
        d['RDN{}'.format(c)] = _k(float, True, True)
        d['VFITOK{}'.format(c)] = _k(bool, False, True)
        for a in range(4):
            d['BIAS{}A{}'.format(c, a)] = _k(float, False, True)
    return d


REDUCTION_CONTRACT = _reduction_contract()


def verify_header(header, htypes=None, dict_head=None, name='header'):
    """blackbox.py:2893-3255 on a header object: every keyword of [dict_head] whose htype is in
    [htypes] must be present when it goes to the database (KeyError), must not be None / 'None'
    unless allowed (ValueError); wrong types and missing non-database keywords only warn.
    -> list of the warnings issued.  dict_head defaults to the reduction keywords this package
    writes (REDUCTION_CONTRACT); a deployment passes the reference's full dictionary."""
    dict_head = REDUCTION_CONTRACT if dict_head is None else dict_head
    htypes_list = [htypes] if isinstance(htypes, str) else list(htypes)
    warnings = []
    for key, e in dict_head.items():
        if e['htype'] not in htypes_list:
            continue
        if key in header:
            v = _val(header, key)
            if e['dtype'] != type(v) and not (isinstance(v, str) and v == 'None'):
                warnings.append('dtype of keyword {}: {} does not match the expected dtype: {} in header of {}'.format(
                    key, type(v), e['dtype'], name))
            if e['DB'] and not e['None_OK'] and (v is None or (isinstance(v, str) and v == 'None')):
                msg = "DataBase keyword {} not allowed to have 'None' or None value in header of {}".format(key, name)
                log.error(msg)
                raise ValueError(msg)
        else:
            msg = 'keyword {} not present in header of {}'.format(key, name)
            if e['DB']:
                log.error(msg)
                raise KeyError(msg)
            warnings.append(msg)
    for w in warnings:
        log.warning(w)
    return warnings
