"""The reference's detector configuration files (``conf/*.by_parts``): ORK pipeline descriptions in YAML whose
``PartsBasedDetector`` pipeline carries the detector's parameters under ``parameters.extra``
(reference cells/detect.cpp:115-126 ``declare_params``: ``model_file`` (required), ``visualize``, ``remove_planes``,
``max_overlap`` = 0.1; conf/config_face.by_parts:31-32 ``model_file`` / ``use_cuda``).

Only the keys that reach the detection path are read; the ROS / ECTO / CouchDB plumbing around them is out of scope.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional

import yaml

from . import filestorage
from .model import Model


@dataclass
class DetectorConfig:
    """what the reference's ECTO cell declares (cells/detect.cpp:115-126)"""
    model_file: str
    max_overlap: float = 0.1          # nonMaximaSuppression overlap used by the callers (cells/detect.cpp:124,238)
    visualize: bool = False
    remove_planes: bool = False
    use_cuda: bool = False            # present in the configs (conf/config_person.by_parts:31), read by nobody in the reference
    pipeline: str = ""


def load_by_parts(path: str) -> List[DetectorConfig]:
    """Every pipeline of type ``PartsBasedDetector`` in a ``.by_parts`` file, in file order."""
    with open(path) as fh:
        doc = yaml.safe_load(fh)
    if not isinstance(doc, dict):
        raise ValueError(f"{path}: not a mapping of pipeline entries")
    out = []
    for name, entry in doc.items():
        if not isinstance(entry, dict) or entry.get("type") != "PartsBasedDetector":
            continue
        params = entry.get("parameters") or {}
        extra = params.get("extra") or {}
        model_file = extra.get("model_file", params.get("model_file"))
        if not model_file:
            raise ValueError(f"{path}: pipeline {name!r} has no model_file (required: cells/detect.cpp:122-123)")
        def pick(key, default):
            return extra.get(key, params.get(key, default))
        out.append(DetectorConfig(model_file=str(model_file), max_overlap=float(pick("max_overlap", 0.1)),
                                  visualize=bool(pick("visualize", False)), remove_planes=bool(pick("remove_planes", False)),
                                  use_cuda=bool(pick("use_cuda", False)), pipeline=str(name)))
    if not out:
        raise ValueError(f"{path}: no pipeline of type PartsBasedDetector")
    return out


def load_model(cfg: DetectorConfig, stand_in: Optional[Model] = None, search_dirs=()) -> Model:
    """The model a configuration names: the file itself, or a file of the same name in `search_dirs` (the
    configs hold absolute paths of the authors' machines), or -- the reference ships no model files: ``models/`` is an
    empty submodule -- the given synthetic stand-in."""
    for cand in [cfg.model_file] + [os.path.join(d, os.path.basename(cfg.model_file)) for d in search_dirs]:
        if os.path.exists(cand):
            return filestorage.deserialize(cand)
    if stand_in is None:
        raise FileNotFoundError(f"model file {cfg.model_file} not found")
    return stand_in
