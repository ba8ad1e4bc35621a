"""N4: the reference's descriptor cache text format (Features.h:62-142)."""
import numpy as np

from stereo_amd import api, featfile, synth


def _image(seed, n, name):
    d = synth.sift_like(seed, max(n, 1))[:n]
    s = 700 + 100 * np.arange(n, dtype=np.float32)
    feats = [api.Feature(scale=1, p=(float(3 * i), float(i) + 0.5), score=float(s[i]), angle=0.25 * i,
                         desc=d[i]) for i in range(n)]
    K = np.array([[994.9, 0, 311.2], [0, 994.9, 255.1], [0, 0, 1]], np.float32)
    return featfile.ImageDescriptor(name, 741, 500, K, np.eye(3, dtype=np.float32), feats)


def test_round_trip_and_format(tmp_path):
    path = str(tmp_path / "desc.dat")
    imgs = [_image(1, 5, "im0.png"), _image(2, 0, "empty.png"), _image(3, 7, "im1.png")]
    assert featfile.SaveImageDescriptorsToFile(path, imgs)
    text = open(path).read()
    assert text.startswith("im0.png 741 500\n994.9 0 311.2 \n")       # operator<< layout, %g floats
    back = featfile.ReadDescriptorsFromFile(path)
    assert [b.filename for b in back] == ["im0.png", "im1.png"]          # images without features are dropped
    assert len(back[0].features) == 5 and len(back[1].features) == 7
    for a, b in zip(imgs[0].features, back[0].features):
        # default ostream precision keeps 6 significant digits
        assert np.allclose(a.desc, b.desc, rtol=1e-5, atol=1e-7)
        assert b.p == (a.p[0], a.p[1]) and abs(b.score - a.score) < 1e-2
    # a second save of what was read is byte-identical (fixed point of the format)
    path2 = str(tmp_path / "desc2.dat")
    featfile.SaveImageDescriptorsToFile(path2, back)
    back2 = featfile.ReadDescriptorsFromFile(path2)
    assert all((x.desc == y.desc).all() for x, y in zip(back[0].features, back2[0].features))
    assert featfile.ReadDescriptorsFromFile(str(tmp_path / "missing.dat")) is None
