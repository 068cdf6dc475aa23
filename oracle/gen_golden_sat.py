"""TEST INFRASTRUCTURE -- golden vectors for the satellite-trail front end (build container only).

    /opt/conda/bin/python3.9 oracle/gen_golden_sat.py

acstools.satdet (the package the reference calls, blackbox.py:4183-4186, `detsat(..., buf=40, sigma=3,
h_thresh=0.2)`) is not in this image; the library functions its published detector is made of are:
numpy.percentile, skimage.exposure.rescale_intensity, skimage.feature.canny,
skimage.morphology.remove_small_objects and a Hough transform (acstools runs the *probabilistic* one,
which draws random pixels; the full accumulator skimage.transform.hough_line is the deterministic
stand-in).  This script runs those functions from the conda environment (scikit-image 0.18.3) on the
seeded scenes of blackbox_amd/synth.py and stores their outputs in tests/golden/sat_front.npz:
percentiles, the rescaled image (checksum), the Canny edge map (sigma 3, thresholds 0.1 / 0.2 of the
maximum), the map after remove_small_objects(60, connectivity 8), and the Hough accumulator over
theta = 2, 2.5, ..., 177.5 degrees (checksum, maximum and its cell); and (mask_fixtures below) the library version of
make_mask + the segments of the real probabilistic Hough transform in tests/golden/sat_mask.npz."""
import hashlib
import json
import os
import sys
import warnings

warnings.filterwarnings('ignore')
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

import numpy as np                                              # noqa: E402
from skimage import exposure, morphology, transform             # noqa: E402
from skimage.feature import canny                               # noqa: E402
from blackbox_amd import synth                                  # noqa: E402  (numpy only)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    out, meta = {}, {}
    for name in synth.SAT_SCENES:
        b = synth.sat_scene(name)
        p1, p2 = np.percentile(b, (4.5, 93.0))
        if p1 < 0:
            p1 = 0.0
        img = exposure.rescale_intensity(b, in_range=(p1, p2))
        immax = np.max(img)
        edge = canny(img, sigma=3, low_threshold=immax * 0.1, high_threshold=immax * 0.2)
        kept = morphology.remove_small_objects(edge, min_size=60, connectivity=8)
        theta = np.radians(np.arange(2, 178, 0.5, dtype=float))
        acc, _, dist = transform.hough_line(kept, theta=theta)
        k = int(np.argmax(acc))
        meta[name] = dict(sha_input=sha(b), p1=float(p1), p2=float(p2), immax=float(immax), sha_rescaled=sha(img),
                          n_edge=int(edge.sum()), n_kept=int(kept.sum()), acc_shape=list(acc.shape), sha_acc=sha(acc.astype(np.int64)),
                          acc_max=int(acc.max()), acc_argmax=[int(k // acc.shape[1]), int(k % acc.shape[1])],
                          rho_offset=int(acc.shape[0] // 2))
        out[name + '_edge'] = np.packbits(edge)
        out[name + '_kept'] = np.packbits(kept)
        print(name, meta[name]['n_edge'], meta[name]['n_kept'], meta[name]['acc_max'], meta[name]['acc_argmax'])
    out['meta'] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'sat_front.npz'), **out)


def mask_fixtures():
    """tests/golden/sat_mask.npz: the back end of acstools.satdet on full scenes.
    * make_mask: the walk of oracle/sattrail.make_mask with every numerical step done by the library --
      skimage.transform.rotate (order 3 there, order 1 back), numpy.median, astropy.stats.sigma_clipped_stats and
      biweight_midvariance -- from the segment the deterministic detector hands over: the mask (bit-packed), every
      window's box, row medians, clipped mean, midvariance and trail rows;
    * the same with the float32 image rotated in float32, as scikit-image 0.18 does when acstools passes it the
      float32 frame: the mask must be the same;
    * skimage.transform.probabilistic_hough_line (acstools' line finder: threshold 210, line_length 200, line_gap 75)
      with seeds 0..9 on the edge map: all segments it returns."""
    sys.path.insert(0, HERE)
    import sattrail as S
    for n, f in {'asscalar': lambda a: a.item(), 'alen': len}.items():      # conda's astropy 4.3.1 predates its numpy 1.26
        if not hasattr(np, n):
            setattr(np, n, f)
    from astropy.stats import biweight_midvariance, sigma_clipped_stats
    out, meta = {}, {}
    for name, p in synth.SAT_MASK_SCENES.items():
        img, truth = synth.sat_full_scene(p['seed'], p['ny'], p['nx'], p['trail'])
        m, nsats, info = S.detect(img)
        assert nsats == 1, (name, info)
        b = S.bin2(img)
        seg = info['segment']

        def rows(medarr, sigma):
            mean = sigma_clipped_stats(medarr)[0]
            var = biweight_midvariance(medarr)
            return np.where(medarr > (mean + (sigma * var)))[0], mean, var
        prims = dict(rotate3=lambda a, deg: transform.rotate(a, deg, resize=True, order=3), rows=rows,
                     rotate1=lambda a, deg: transform.rotate(a, deg, resize=True, order=1))
        mask, dbg = S.make_mask(b, seg, return_debug=True, prims=prims)
        prims32 = dict(prims, rotate3=lambda a, deg: transform.rotate(a.astype(np.float32), deg, resize=True, order=3).astype(np.float64))
        mask32 = S.make_mask(b, seg, prims=prims32)
        theta = np.radians(np.arange(2, 178, 0.5, dtype=float))
        edge = S.edges(b)
        segs = []
        for seed in range(10):
            r = transform.probabilistic_hough_line(edge, threshold=210, line_length=200, line_gap=75, theta=theta, seed=seed)
            segs += [[a[0], a[1], c[0], c[1], seed] for a, c in r]
        assert segs, name
        w = dbg['windows']
        meta[name] = dict(sha_input=sha(img), segment=seg, deg=float(dbg['deg']), start=[float(v) for v in dbg['start']],
                          rot_shape=list(dbg['rot_shape']), nwin=len(w), boxes=[list(x['box']) for x in w], z=[x['z'] for x in w],
                          n_mask=int(mask.sum()), same_with_float32_rotation=bool(np.array_equal(mask, mask32)),
                          n_diff_float32=int((mask != mask32).sum()))
        out[name + '_mask'] = np.packbits(mask)
        out[name + '_medarr'] = np.concatenate([x['medarr'] for x in w])
        out[name + '_mean'] = np.array([x['mean'] for x in w])
        out[name + '_var'] = np.array([x['var'] for x in w])
        out[name + '_pht'] = np.array(segs, np.int32)
        print(name, seg, 'windows', len(w), 'mask px', int(mask.sum()), 'float32 rotation: same' if meta[name]['same_with_float32_rotation']
              else 'float32 rotation: %d px differ' % meta[name]['n_diff_float32'], 'PHT segments', len(segs))
    out['meta'] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'sat_mask.npz'), **out)


if __name__ == '__main__':
    if 'front' in sys.argv[1:] or len(sys.argv) == 1:
        main()
    if 'mask' in sys.argv[1:] or len(sys.argv) == 1:
        mask_fixtures()
