"""Writes the small rasters textured.xml refers to (procedural, seeded): checker.png (8x8 RGB), gray.png (16x16 L), bumps.png
(32x32 RGB tangent-space normal map). Run from this directory; the PNGs are committed."""
import importlib
import os
import sys

from PIL import Image

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..")))
S = importlib.import_module("nano-kazen_amd").scenes
chk, noise, gray, nrm = S._test_images()
here = os.path.dirname(os.path.abspath(__file__))
Image.fromarray(chk, "RGB").save(os.path.join(here, "checker.png"))
Image.fromarray((gray[:, :, 0] * 255).astype("uint8"), "L").save(os.path.join(here, "gray.png"))
Image.fromarray(nrm, "RGB").save(os.path.join(here, "bumps.png"))
