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
theta = 2, 2.5, ..., 177.5 degrees (checksum, maximum and its cell)."""
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


if __name__ == '__main__':
    main()
