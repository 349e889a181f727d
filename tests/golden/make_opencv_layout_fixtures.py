"""Writes tests/golden/toy_model_opencv_layout.{yml,xml} and ..._filters.npy: a small model document typed out in the
layout OpenCV's cv::FileStorage writer produces (YAML flow sequences wrapped over lines, `!!opencv-matrix` nodes; XML
`<_>` items, `type_id="opencv-matrix"`, wrapped number runs, an empty `<defid>` for the root), independently of this
repository's own writers (`filestorage.serialize*`).  No file written by a real OpenCV ships with the reference, so
the readers stay "parity unpinned"; these documents pin the two readers (Python, C++) to each other and to the layout.
Run from the repository root:  python tests/golden/make_opencv_layout_fixtures.py
"""
import numpy as np

rng = np.random.default_rng(123)
f = [np.round(rng.standard_normal((3, 96)) * 0.05, 6) for _ in range(3)]   # 3 filters of 3x3x32


def yml_data(a):
    toks = ["%.8e" % v for v in a.ravel()]
    lines, cur = [], "      data: [ "
    for i, t in enumerate(toks):
        piece = t + (", " if i + 1 < len(toks) else " ]")
        if len(cur) + len(piece) > 76:
            lines.append(cur.rstrip())
            cur = "         "
        cur += piece
    lines.append(cur)
    return "\n".join(lines)


def xml_data(a, ind):
    toks = ["%.8e" % v for v in a.ravel()]
    return "\n".join(" " * ind + " ".join(toks[i:i + 4]) for i in range(0, len(toks), 4))


yml = ["%YAML:1.0", 'name: "toy_opencv_layout"', "interval: 4", "thresh: -7.5000000000000000e-01", "sbin: 8", "norient: 18",
       "flen: 32", "filtersw:"]
for a in f:
    yml += ["   - !!opencv-matrix", "      rows: 3", "      cols: 96", "      dt: d", yml_data(a)]
yml += ["biasw: [ 1.00000001e-01, -2.00000003e-01, 3.00000012e-01,", "    -4.00000006e-01, 5.00000000e-01 ]",
        "anchors: [ 1, -2, 0, 3 ]",
        "defs:", "   - [ 9.99999978e-03, 0., 1.99999996e-02, 1.00000005e-03 ]", "   - [ 2.99999993e-02, -1.00000005e-03, 9.99999978e-03, 0. ]",
        "indexers:", "   component-0:", "      part-0:", "         parentid: -1", "         filterid: [ 0 ]", "         biasid: [ 0 ]",
        "         defid: [ ]",
        "      part-1:", "         parentid: 0", "         filterid: [ 1, 2 ]", "         biasid: [ 1, 3 ]", "         defid: [ 0, 1 ]"]
open("tests/golden/toy_model_opencv_layout.yml", "w").write("\n".join(yml) + "\n")

xml = ['<?xml version="1.0"?>', "<opencv_storage>", "<name>toy_opencv_layout</name>", "<interval>4</interval>",
       "<thresh>-7.5000000000000000e-01</thresh>", "<sbin>8</sbin>", "<norient>18</norient>", "<flen>32</flen>", "<filtersw>"]
for a in f:
    xml += ['  <_ type_id="opencv-matrix">', "    <rows>3</rows>", "    <cols>96</cols>", "    <dt>d</dt>", "    <data>",
            xml_data(a, 6) + "</data></_>"]
xml[-1] += "</filtersw>"
xml += ["<biasw>", "  1.00000001e-01 -2.00000003e-01 3.00000012e-01 -4.00000006e-01", "  5.00000000e-01</biasw>",
        "<anchors>", "  1 -2 0 3</anchors>", "<defs>", "  <_>", "    9.99999978e-03 0. 1.99999996e-02 1.00000005e-03</_>", "  <_>",
        "    2.99999993e-02 -1.00000005e-03 9.99999978e-03 0.</_></defs>", "<indexers>", "  <component-0>", "    <part-0>",
        "      <parentid>-1</parentid>", "      <filterid>", "        0</filterid>", "      <biasid>", "        0</biasid>",
        "      <defid></defid></part-0>", "    <part-1>", "      <parentid>0</parentid>", "      <filterid>", "        1 2</filterid>",
        "      <biasid>", "        1 3</biasid>", "      <defid>", "        0 1</defid></part-1></component-0></indexers>",
        "</opencv_storage>"]
open("tests/golden/toy_model_opencv_layout.xml", "w").write("\n".join(xml) + "\n")
np.save("tests/golden/toy_model_opencv_layout_filters.npy", np.stack(f))
