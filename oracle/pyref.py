"""ctypes loader for oracle/_ref/libjuce_colour_ref.so -- the reference's own juce::Colour::withRotatedHue, compiled from the sources
under /root/reference (oracle/ref_juce_colour.cpp; `make -C oracle _ref`).  Container-only test infrastructure: the reference is not on
the GPU box, so nothing marked gpu, smoke() or bench.py may need this module; the CPU tests that use it skip when it is unavailable,
and the vectors it generated are committed under tests/golden/ (tools/make_golden.py)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_ref", "libjuce_colour_ref.so")
REFERENCE = "/root/reference/JuceLibraryCode/modules/juce_graphics/colour/juce_Colour.cpp"
_lib = None


def build() -> bool:
    """(Re)build the shared object if the reference is here; False when it is not (then nothing is built)."""
    if not os.path.exists(REFERENCE):
        return False
    subprocess.check_call(["make", "-s", "-C", HERE, "_ref"])
    return os.path.exists(LIB)


def available() -> bool:
    return os.path.exists(LIB) or build()


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref/libjuce_colour_ref.so: the reference's sources are not in this environment")
        _lib = C.CDLL(LIB, mode=os.RTLD_LAZY)      # (juce::String's unresolved members are never called)
        _lib.sgzref_rotate_hue_rgb8.argtypes = [C.c_void_p, C.c_float, C.c_void_p]
        _lib.sgzref_rotate_hue_rgb8.restype = None
        _lib.sgzref_colour_rotation_rgb8.argtypes = [C.c_void_p, C.c_uint64, C.c_float, C.c_int, C.c_void_p]
        _lib.sgzref_colour_rotation_rgb8.restype = None
    return _lib


def rotate_hue(rgb, amount: float) -> np.ndarray:
    """juce::Colour(r, g, b).withRotatedHue(amount) -> rgb (juce_Colour.cpp:331-336)"""
    a = np.ascontiguousarray(rgb, np.uint8)
    out = np.zeros(3, np.uint8)
    lib().sgzref_rotate_hue_rgb8(a.ctypes.data_as(C.c_void_p), C.c_float(amount), out.ctypes.data_as(C.c_void_p))
    return out


def colour_rotation(base, index: int, size: float, stereo: bool = False) -> np.ndarray:
    """Signalizer::ColourRotation(base, size, stereo)[index] (CommonSignalizer.h:921-937)"""
    a = np.ascontiguousarray(base, np.uint8)
    out = np.zeros(3, np.uint8)
    lib().sgzref_colour_rotation_rgb8(a.ctypes.data_as(C.c_void_p), C.c_uint64(index), C.c_float(size), int(bool(stereo)), out.ctypes.data_as(C.c_void_p))
    return out


def spectrogram_colours(colours, pairs: int, rotation: int) -> np.ndarray:
    """TransformConstant::generateSpectrogramColourRotation(rotation) (TransformConstant.h:55-65) for colourSpecs built as in
    Spectrum.cpp:398-402 (ColourRotation(colour, pairs, false)): [6][3] RGB8"""
    out = np.zeros((len(colours), 3), np.uint8)
    for i, c in enumerate(colours):
        out[i] = colour_rotation(c, 0 if i == 0 else rotation, float(pairs), False)
    return out
