"""Python restatement of the reference OBJ loader's semantics (test oracle for
gpt_obj_load).  Follows objcore/objloader.cpp:18-122 (line dispatch, ' '-separated
tokens, atof, index = atof(token) - 1 truncated to int, three face syntaxes) and
PathTracer::prepMesh (path_tracer.cpp:52-69: first three vertices of each face,
normal = the normal at the face's FIRST normal index, not renormalised)."""
import numpy as np

f32 = np.float32


def _atof(tok):
    # C atof: longest valid numeric prefix, 0.0 when there is none
    tok = tok.strip(" \t")
    best = 0.0
    for end in range(len(tok), 0, -1):
        try:
            best = float(tok[:end])
            break
        except ValueError:
            continue
    return best


def load_obj(path):
    points, normals, faces, facenormals = [], [], [], []
    with open(path, "r") as fh:
        text = fh.read()
    for line in text.split("\n"):
        if line == "":
            line = "42"
        toks = line.split(" ")
        if line[0] == "v" and len(line) > 1 and line[1] == "t":
            continue
        elif line[0] == "v" and len(line) > 1 and line[1] == "n":
            t = (toks + ["", "", ""])[1:4]
            normals.append([f32(_atof(x)) for x in t])
        elif line[0] == "v":
            t = (toks + ["", "", ""])[1:4]
            points.append([f32(_atof(x)) for x in t])
        elif line[0] == "f":
            verts = toks[1:]
            if "//" in line:
                pl, nl = [], []
                for v in verts:
                    parts = v.split("/")
                    pl.append(int(_atof(parts[0]) - 1))
                    nl.append(int(_atof(parts[2] if len(parts) > 2 else "") - 1))
                faces.append(pl)
                facenormals.append(nl)
            elif "/" in line:
                pl, nl = [], []
                for v in verts:
                    parts = v.split("/")
                    pl.append(int(_atof(parts[0]) - 1))
                    if len(parts) > 2:
                        nl.append(int(_atof(parts[2]) - 1))
                faces.append(pl)
                facenormals.append(nl)
            else:
                faces.append([int(_atof(v) - 1) for v in verts])
    return (np.array(points, dtype=f32).reshape(-1, 3), np.array(normals, dtype=f32).reshape(-1, 3),
            faces, facenormals)


def triangles(path):
    """(verts[n,9], normals[n,3]) as prepMesh would fill Poly.p0/p1/p2/n."""
    pts, nrm, faces, fn = load_obj(path)
    n = len(faces)
    verts = np.zeros((n, 9), dtype=f32)
    normals = np.zeros((n, 3), dtype=f32)
    for i, fc in enumerate(faces):
        verts[i, 0:3] = pts[fc[0]]
        verts[i, 3:6] = pts[fc[1]]
        verts[i, 6:9] = pts[fc[2]]
        if i < len(fn) and len(fn[i]) > 0:
            normals[i] = nrm[fn[i][0]]
        else:
            a = verts[i, 3:6] - verts[i, 0:3]
            b = verts[i, 6:9] - verts[i, 0:3]
            c = np.cross(a, b).astype(f32)
            ln = f32(np.sqrt(f32(c[0] * c[0] + c[1] * c[1] + c[2] * c[2])))
            normals[i] = c / ln if ln > 0 else c
    bbox = np.concatenate([pts.min(axis=0), pts.max(axis=0)]) if len(pts) else np.zeros(6, f32)
    return verts, normals, bbox
