"""Fixture generator (runs in the build container only, where /root/reference exists): BASELINE.json configs[0] is "3-view
sora/50 iters plumbing" on the reference's own example images, reference assets/sora/{Art,Santorini}/images.  The GPU box has no
/root/reference, so the three `Art` frames (1280 x 720) are committed as test DATA under tests/golden/sora_art/ — decoded and
re-encoded as JPEG (quality 90) by this script, at their native size: the same pictures to within JPEG noise, not the same files.  Nothing else of the scene exists offline: the reference
obtains points and poses from MASt3R (init_geo.py), which needs its checkpoint; tests/sora_util.py replaces that stage by a
synthetic pointmap coloured from the frames and poses on an arc, and says so wherever the result is quoted.

usage: python tests/golden/make_golden_sora.py"""
import hashlib
import os
import shutil

SRC = "/root/reference/assets/sora/Art/images"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sora_art")

if __name__ == "__main__":
    os.makedirs(DST, exist_ok=True)
    lines = []
    from PIL import Image
    for name in sorted(os.listdir(SRC)):
        out = "art_frame_" + name
        with Image.open(os.path.join(SRC, name)) as im:
            im.convert("RGB").save(os.path.join(DST, out), "JPEG", quality=90)
        with open(os.path.join(DST, out), "rb") as fh:
            lines.append(f"{hashlib.sha256(fh.read()).hexdigest()}  {out}")
    with open(os.path.join(DST, "SHA256SUMS"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))
