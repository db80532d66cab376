"""Derive the name->index tables of an imitation-learning robot from its MJCF: what
mushroom-rl's ObservationHelper / MuJoCo constructor obtain from the compiled model
(joint qpos/dof addresses, joint ranges, actuator order, ctrlrange, gear).  A plain
ElementTree walk in document order - dm_control / mujoco are not needed."""
import xml.etree.ElementTree as ET

import numpy as np


def tables_from_mjcf(xml_path, removed_joints=(), removed_motors=()):
    """Returns dict(joints=[(name, qposadr, dofadr, lo, hi)], motors=[(name, joint, gear, lo, hi)],
    nq, nv).  free joints take 7/6 addresses, ball 4/3, slide/hinge 1/1."""
    root = ET.parse(xml_path).getroot()
    joints, qadr, vadr = [], 0, 0

    def walk(body):
        nonlocal qadr, vadr
        for el in body:
            if el.tag == "freejoint" or (el.tag == "joint" and el.get("type") == "free"):
                joints.append((el.get("name"), qadr, vadr, -np.inf, np.inf))
                qadr, vadr = qadr + 7, vadr + 6
            elif el.tag == "joint":
                if el.get("name") in removed_joints:
                    continue
                nq, nv = (4, 3) if el.get("type") == "ball" else (1, 1)
                rng = el.get("range")
                lo, hi = (float(v) for v in rng.split()) if rng else (-np.inf, np.inf)
                joints.append((el.get("name"), qadr, vadr, lo, hi))
                qadr, vadr = qadr + nq, vadr + nv
            elif el.tag == "body":
                walk(el)

    walk(root.find("worldbody"))
    default_ctrl = (-np.inf, np.inf)
    for d in root.iter("default"):
        for m in d.findall("motor"):
            if m.get("ctrlrange"):
                default_ctrl = tuple(float(v) for v in m.get("ctrlrange").split())
    motors = []
    act = root.find("actuator")
    for m in (act if act is not None else []):
        if m.get("name") in removed_motors:
            continue
        cr = tuple(float(v) for v in m.get("ctrlrange").split()) if m.get("ctrlrange") else default_ctrl
        motors.append((m.get("name"), m.get("joint"), float((m.get("gear") or "1").split()[0]), cr[0], cr[1]))
    return dict(joints=joints, motors=motors, nq=qadr, nv=vadr)


def geoms_from_mjcf(xml_path):
    """Geom names (None if unnamed) in compiled geom-id order: bodies are numbered depth-first
    in document order (world = 0) and a body's own geoms come before those of later bodies."""
    root = ET.parse(xml_path).getroot()
    bodies = []

    def walk(body):
        bodies.append(body)
        for el in body:
            if el.tag == "body":
                walk(el)

    walk(root.find("worldbody"))
    return [g.get("name") for b in bodies for g in b if g.tag == "geom"]


def bodies_from_mjcf(xml_path):
    """(body names in compiled body-id order, world first; geom -> body id list in geom-id order)."""
    root = ET.parse(xml_path).getroot()
    bodies = []

    def walk(body):
        bodies.append(body)
        for el in body:
            if el.tag == "body":
                walk(el)

    walk(root.find("worldbody"))
    names = ["world"] + [b.get("name") for b in bodies[1:]]
    geom_body = [i for i, b in enumerate(bodies) for g in b if g.tag == "geom"]
    return names, geom_body
