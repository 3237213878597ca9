"""OBJ / MTL loading with the entry points of /root/reference/mesh_parser.py
(`read_mesh` :7, `MaterialFactory` :34, `read_mat` :90, `read_mesh_complex` :94).

These run once at scene-build time on the host.  They reproduce the reference's
reading of the files, including what is peculiar to it:

* `read_mesh_complex` closes a group when the *next vertex line* arrives after a `g`
  line, so faces listed before their `g` line belong to that later group, and the last
  group is closed at end of file (:114-118, :134-135);
* negative face indices count back from the vertices read so far (:123);
* every mesh of a file shares the one growing vertex list (:95, :106);
* the MTL reader takes `Ka` as the surface colour and also as `ka`, and reduces `Kd`
  and `Ks` to scalars `Kd[0]/Ka[0]`, `Ks[0]/Ka[0]` (:47-59).
"""
import os

import numpy as np

from geometry import Mesh
from material import Matte, Phong, Emissive


def _fields(line):
    return line.split()


def read_mesh(f, face_limit=None):
    """Vertices from `v` lines, 1-based faces from `f` lines; everything else ignored."""
    vertices, faces = [], []
    for line in f:
        tok = _fields(line)
        if not tok:
            continue
        if tok[0] == 'v':
            vertices.append(np.array([float(t) for t in tok[1:]]))
        elif tok[0] == 'f':
            faces.append([int(t) - 1 for t in tok[1:]])
    if face_limit:
        faces = faces[:face_limit]
    return Mesh(vertices, faces)


class MaterialFactory(object):
    def __init__(self, f, mat_sampler):
        self.materials = {}
        name, params = None, {}
        for line in f:
            tok = _fields(line)
            if '#' in tok:
                tok = tok[:tok.index('#')]
            if not tok:
                continue
            if tok[0] == 'newmtl':
                if name is not None:
                    self._close(name, params, mat_sampler)
                    params = {}
                name = tok[1]
            else:
                params[tok[0]] = np.array([float(t) for t in tok[1:]])
        if params:
            self._close(name, params, mat_sampler)

    def _close(self, name, params, mat_sampler):
        Ka = params.get('Ka')
        Kd, Ks = params.get('Kd'), params.get('Ks')
        Ke, Ns = params.get('Ke'), params.get('Ns')
        color = Ka
        if 'Kd' in params:
            Kd = Kd[0] / color[0]
        if 'Ks' in params:
            Ks = Ks[0] / color[0]
        mat = None
        if int(params.get('illum')[0]) == 2:
            if 'Ke' in params and sum(Ke):
                mat = Emissive(1., Ke)
            elif Ns is not None and Ns[0] != 0.:
                mat = Matte(ka=Ka, kd=Kd, cd=color, sampler=mat_sampler)
            elif Ns is not None:
                mat = Phong(ka=Ka, kd=Kd, ks=Ks, cd=color, exp=Ns[0])
        if mat is None:
            raise ValueError("material %r: only illum 2 materials are understood" % name)
        self.materials[name] = mat

    def get_material(self, name):
        return self.materials[name]


def read_mat(f, mat_sampler):
    return MaterialFactory(f, mat_sampler)


class MeshGroups(dict):
    """{group name: Mesh} in file order that also answers the Python 2 iteration methods the
    reference's scene files call on it (buildfunctions/cornellbox.py:44 `iteritems`).  A
    Python 2 dict iterates in hash order; the order here is the file's, as Python 3 gives."""

    def iteritems(self):
        return iter(self.items())

    def itervalues(self):
        return iter(self.values())

    def iterkeys(self):
        return iter(self.keys())

    def has_key(self, key):
        return key in self


def read_mesh_complex(path, mat_sampler):
    """Grouped OBJ with an MTL library -> {group name: Mesh} in file order."""
    folder = os.path.dirname(os.path.realpath(path))
    vertices, faces, meshes = [], [], MeshGroups()
    group = material = library = None

    def close():
        meshes[group] = Mesh(vertices, faces, library.get_material(material))

    with open(path) as f:
        for line in f:
            tok = _fields(line)
            if not tok:
                continue
            key = tok[0]
            if key == 'v':
                if group is not None:
                    close()
                    faces, group = [], None
                vertices.append(np.array([float(t) for t in tok[1:]]))
            elif key == 'f':
                faces.append([int(t) - 1 if int(t) > 0 else len(vertices) + int(t) for t in tok[1:]])
            elif key == 'g':
                group = tok[1]
            elif key == 'usemtl':
                material = tok[1]
            elif key == 'mtllib':
                with open(os.path.join(folder, tok[1])) as lib:
                    library = read_mat(lib, mat_sampler)
    if faces:
        close()
    return meshes
