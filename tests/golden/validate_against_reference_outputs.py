"""Broad check of the oracle against ALL outputs the reference pipeline left in
/root/reference/imagenette2/transformed (run in the build container; the GPU box has no
reference).  For every image that has an identity-parameter output (the original after one
JPEG round trip) every other transformation is applied to that proxy with the parameter in
the file name and compared with the reference's file.  Writes reference_outputs_summary.tsv
(per transformation: cases, size mismatches, min / median PSNR after a JPEG round trip).
The 30 files used by tests/test_reference_outputs.py are a subset."""
import collections, io, os, sys
import numpy as np
from PIL import Image
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import imgxf_oracle as O

D = "/root/reference/imagenette2/transformed"
T = ("lighten_darken", "gaussian_noise", "translation", "rotation", "contrast", "scale", "shear", "blur")
IDENT = {"contrast": "1.0", "gaussian_noise": "0.0", "rotation": "0.0", "scale": "1.0", "lighten_darken": "0.0", "blur": "0.0"}
FN = {"lighten_darken": O.apply_brightness, "rotation": O.apply_rotation, "contrast": O.apply_contrast,
      "scale": O.apply_scale, "shear": O.apply_shear, "blur": O.apply_blur}

def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)

def jpeg(a):
    buf = io.BytesIO(); Image.fromarray(a).save(buf, format="JPEG")
    return np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))

groups = collections.defaultdict(dict)
for f in os.listdir(D):
    if not f.endswith("_corrupted.JPEG"): continue
    stem = f[:-len("_corrupted.JPEG")]
    for t in T:
        i = stem.find("_" + t + "_")
        if i > 0:
            groups[stem[:i]][t] = (stem[i + len(t) + 2:], f); break

stats = collections.defaultdict(list)
fixed_vs_float = []          # (radius, psnr float definition, psnr OpenCV fixed-point restatement, max |diff|, fraction differing)
mism = collections.Counter()
for name, ts in sorted(groups.items()):
    ids = [t for t, (v, _) in ts.items() if IDENT.get(t) == v]
    if not ids: continue
    proxy = np.asarray(Image.open(os.path.join(D, ts[ids[0]][1])).convert("RGB"))
    for t, (v, f) in ts.items():
        if t in ids or t == "gaussian_noise": continue
        ref = np.asarray(Image.open(os.path.join(D, f)).convert("RGB"))
        if t == "translation":
            tx, ty = v.split("_"); out = O.apply_translation(proxy, float(tx), float(ty))
        else:
            out = FN[t](proxy, float(v))
        if out.shape != ref.shape:
            mism[t] += 1; continue
        stats[t].append(psnr(jpeg(out), ref))
        if t == "blur" and float(v) > 0:
            fx = O.gaussian_blur_cv_fixed(proxy, O.blur_ksize(float(v)), float(v))
            fixed_vs_float.append((float(v), stats[t][-1], psnr(jpeg(fx), ref),
                                   int(np.abs(fx.astype(int) - out).max()), float((fx != out).mean())))
with open(os.path.join(HERE, "reference_outputs_summary.tsv"), "w") as fh:
    fh.write("transformation\tcases\tsize_mismatches\tmin_psnr_db\tmedian_psnr_db\n")
    for t in sorted(stats):
        v = np.array(stats[t]); line = f"{t}\t{len(v)}\t{mism[t]}\t{v.min():.1f}\t{np.median(v):.1f}"
        print(line); fh.write(line + "\n")
a = np.array(fixed_vs_float)
with open(os.path.join(HERE, "reference_outputs_blur_fixed_vs_float.tsv"), "w") as fh:
    fh.write("radius\tpsnr_float_definition_db\tpsnr_cv_fixed_point_db\tmax_abs_diff\tfraction_differing\n")
    for row in sorted(fixed_vs_float):
        fh.write("%.1f\t%.2f\t%.2f\t%d\t%.4f\n" % row)
print("blur: float definition %.2f dB, OpenCV fixed-point restatement %.2f dB on average; fixed-point better in %d of %d"
      % (a[:, 1].mean(), a[:, 2].mean(), int((a[:, 2] > a[:, 1]).sum()), len(a)))
