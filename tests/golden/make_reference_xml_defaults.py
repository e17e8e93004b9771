#!/usr/bin/env python3
"""Parses the parameter files the reference ships -- resources/putslammatcherOpenCVParameters.xml (RANSAC element :29-37,
MatcherOpenCV element :64-78), resources/putslammatcherOpenCVParametersLC.xml (:30) and
resources/datasetConfig/freiburg1_desk.xml (:5-8, :20) -- and writes the values this repository hard-codes as defaults to
tests/golden/reference_xml_defaults.json.  Data only (numbers read from the reference's resource files), generated in the build
container where /root/reference exists; tests/test_reference_defaults.py compares the product's defaults with it on any box and
re-parses the XML where the reference is present."""
import json
import os
import sys
import xml.etree.ElementTree as ET

REF = os.environ.get("PUTSLAM_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def _num(v):
    f = float(v)
    return int(f) if f == int(f) and "." not in v and "e" not in v.lower() else f


def parse(ref=REF):
    out = {}
    for key, rel in (("vo", "resources/putslammatcherOpenCVParameters.xml"), ("lc", "resources/putslammatcherOpenCVParametersLC.xml")):
        root = ET.parse(os.path.join(ref, rel)).getroot()
        ransac = root.find(".//RANSAC")
        out[key] = {"file": rel, "RANSAC": {k: _num(v) for k, v in ransac.attrib.items()}}
        m = root.find(".//MatcherOpenCV")
        out[key]["MatcherOpenCV"] = {k: _num(m.attrib[k]) for k in ("matchingXYZSphereRadius", "matchingXYZacceptRatioOfBestMatch")}
    rel = "resources/datasetConfig/freiburg1_desk.xml"
    # (the file has two top-level elements: Model and datasetPath -- wrap them to make it one document)
    txt = open(os.path.join(ref, rel)).read()
    body = txt[txt.index("?>") + 2:] if txt.lstrip().startswith("<?xml") else txt
    root = ET.fromstring("<r>" + body + "</r>")
    fl, fa, di, sz = root.find(".//focalLength"), root.find(".//focalAxis"), root.find(".//rgbDistortion"), root.find(".//imageSize")
    dp = root.find(".//datasetPath")
    out["camera"] = {"file": rel, "fu": float(fl.attrib["fu"]), "fv": float(fl.attrib["fv"]), "Cu": float(fa.attrib["Cu"]), "Cv": float(fa.attrib["Cv"]),
                     "rgbDistortion": [float(di.attrib[k]) for k in ("k1", "k2", "p1", "p2", "k3")],
                     "sizeU": int(sz.attrib["sizeU"]), "sizeV": int(sz.attrib["sizeV"]), "depthImageScale": float(dp.attrib["depthImageScale"])}
    return out


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("no reference checkout at %s" % REF)
    with open(os.path.join(HERE, "reference_xml_defaults.json"), "w") as f:
        json.dump(parse(), f, indent=1, sort_keys=True)
        f.write("\n")
    print(open(os.path.join(HERE, "reference_xml_defaults.json")).read())
