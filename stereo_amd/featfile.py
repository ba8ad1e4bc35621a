"""The reference's descriptor cache file: whitespace text written by
operator<< of ImageDescriptor / Feature (stereo/Features.h:62-83, 95-142) through
SaveImageDescriptorsToFile / ReadDescriptorsFromFile (stereo/Features.cpp:1219-1276).

Lets feature dumps produced by the reference program be replayed through
MatchDescriptors here, and vice versa.  Host-side IO only.
"""
from dataclasses import dataclass, field

import numpy as np

from .api import Feature, DESC_LENGTH


@dataclass
class ImageDescriptor:
    """struct ImageDescriptor, stereo/Features.h:86-143."""
    filename: str = ""
    width: int = 0
    height: int = 0
    K: np.ndarray = field(default_factory=lambda: np.zeros((3, 3), np.float32))
    E: np.ndarray = field(default_factory=lambda: np.zeros((3, 3), np.float32))
    features: list = field(default_factory=list)


def _g(x):
    # C++ ostream default float formatting: %g with 6 significant digits
    return "%g" % float(x)


def _feature_text(f):
    # Features.h:62-70
    head = [str(int(f.scale)), _g(f.p[0]), _g(f.p[1]), _g(f.score), _g(f.angle),
            _g(f.distFromBestMatch), _g(f.depth)]
    return " ".join(head + [_g(v) for v in np.asarray(f.desc, np.float32)])


def SaveImageDescriptorsToFile(filename, images):
    """Features.cpp:1219-1237.  Returns False if the file cannot be opened."""
    try:
        fh = open(filename, "w")
    except OSError:
        return False
    with fh:
        for im in images:  # Features.h:95-116
            fh.write("%s %d %d\n" % (im.filename, im.width, im.height))
            for M in (im.K, im.E):
                for r in range(3):
                    fh.write("".join(_g(M[r][c]) + " " for c in range(3)) + "\n")
            fh.write("%d\n" % len(im.features))
            for f in im.features:
                fh.write(_feature_text(f) + " ")
    return True


def ReadDescriptorsFromFile(filename):
    """Features.cpp:1242-1276.  Returns the list of images (those with at least
    one feature, :1262-1266), or None if the file cannot be opened."""
    try:
        tokens = open(filename).read().split()
    except OSError:
        return None
    images, pos = [], 0
    per_feature = 7 + DESC_LENGTH
    while pos < len(tokens):
        try:  # Features.h:117-142
            name, width, height = tokens[pos], int(tokens[pos + 1]), int(tokens[pos + 2])
            pos += 3
            K = np.array(tokens[pos:pos + 9], np.float32).reshape(3, 3)
            E = np.array(tokens[pos + 9:pos + 18], np.float32).reshape(3, 3)
            n = int(tokens[pos + 18])
            pos += 19
            vals = np.array(tokens[pos:pos + n * per_feature], np.float64)
            if len(vals) != n * per_feature:
                break  # truncated file: the reference's stream goes bad and its loop ends
            pos += n * per_feature
        except (ValueError, IndexError):
            break
        feats = []
        for row in vals.reshape(n, per_feature):
            feats.append(Feature(scale=int(row[0]), p=(float(np.float32(row[1])), float(np.float32(row[2]))),
                                 score=float(np.float32(row[3])), angle=float(np.float32(row[4])),
                                 desc=row[7:].astype(np.float32), distFromBestMatch=float(np.float32(row[5])),
                                 depth=float(np.float32(row[6]))))
        if n >= 1:
            images.append(ImageDescriptor(name, width, height, K, E, feats))
    return images
