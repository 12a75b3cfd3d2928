"""Mints tests/golden/s3_frames/*.png: frames 90..97 of the reference's sample sequence
(experiments/s3/costado_recto1, the frames samples/EKF/main.cpp starts on), reduced to 320x240 gray so that the GPU
box -- which has no /root/reference -- can run the image-in pipeline on real imagery.  Data only (input vectors);
the expected outputs are computed by the oracle at test time.  Run in the build container:
    python tests/golden/make_s3_frames.py"""
import os

from PIL import Image

SRC = "/root/reference/experiments/s3/costado_recto1"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "s3_frames")

if __name__ == "__main__":
    os.makedirs(DST, exist_ok=True)
    for k, i in enumerate(range(90, 98)):
        im = Image.open(f"{SRC}/{i:05d}.png").convert("L").resize((320, 240), Image.BOX)
        im.save(os.path.join(DST, f"{k:05d}.png"), optimize=True)
    print(sorted(os.listdir(DST)))
