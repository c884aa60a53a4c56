"""Case (reference grid) loading and the static tables the load-flow engine needs.

Mirrors what the reference obtains from ``pypower.api.loadcase`` inside ``Grid.__init__``
(reference: pypownet/grid.py:60-93) and the id conventions of the doubled grid
(pypownet/__init__.py:10, parameters/make_reference_grid.py:39-46): every substation ``i`` owns two
busbars, the real bus row ``i`` and an artificial twin ``"666"+str(i)`` stored in the second half of
``bus``.  All tables below are *index* based (0..nS-1 substations, node bit 0/1) so that no string
id arithmetic is needed at run time.
"""
import json
import os

import numpy as np

ARTIFICIAL_NODE_STARTING_STRING = '666'  # reference: pypownet/__init__.py:10

# MATPOWER v2 column indices (0-based)
BUS_I, BUS_TYPE, PD, QD, GS, BS, BUS_AREA, VM, VA, BASE_KV = range(10)
GEN_BUS, PG, QG, QMAX, QMIN, VG, MBASE, GEN_STATUS = range(8)
F_BUS, T_BUS, BR_R, BR_X, BR_B, RATE_A, RATE_B, RATE_C, TAP, SHIFT, BR_STATUS = range(11)
PF, QF, PT, QT = 13, 14, 15, 16


def load_case_file(path):
    """Return the case dict {'version','baseMVA','bus','gen','branch'} with float64 arrays.

    ``*.py``: PYPOWER case format: a bare function named after the file's basename, evaluated with numpy's
    ``array`` in scope (same contract as ``pypower.loadcase``; reference call site pypownet/grid.py:65).
    ``*.json``: the same arrays as plain JSON lists (format of this repository's fixtures).
    """
    if path.endswith('.json'):
        with open(path, 'r') as f:
            raw = json.load(f)
        return {'version': str(raw.get('version', '2')), 'baseMVA': float(raw['baseMVA']),
                'bus': np.asarray(raw['bus'], dtype=np.float64),
                'gen': np.asarray(raw['gen'], dtype=np.float64),
                'branch': np.asarray(raw['branch'], dtype=np.float64)}
    if path.endswith('.py'):
        with open(path, 'r') as f:
            src = f.read()
        scope = {'array': np.array}
        exec(compile(src, path, 'exec'), scope)
        fname = os.path.splitext(os.path.basename(path))[0]
        ppc = scope[fname]()
        return {'version': str(ppc.get('version', '2')), 'baseMVA': float(ppc['baseMVA']),
                'bus': np.asarray(ppc['bus'], dtype=np.float64),
                'gen': np.asarray(ppc['gen'], dtype=np.float64),
                'branch': np.asarray(ppc['branch'], dtype=np.float64)}
    raise ValueError('unsupported case file %s (expected .py or .json)' % path)


def save_case_json(ppc, path):
    with open(path, 'w') as f:
        json.dump({'version': str(ppc.get('version', '2')), 'baseMVA': float(ppc['baseMVA']),
                   'bus': np.asarray(ppc['bus']).tolist(), 'gen': np.asarray(ppc['gen']).tolist(),
                   'branch': np.asarray(ppc['branch']).tolist()}, f)


def save_case_py(ppc, path):
    """The PYPOWER case format the reference's tooling writes (``savecase(output_file, mpc)``, parameters/make_reference_grid.py:63)
    and its environments ship (``parameters/<env>/<level>/reference_grid.py``, read back by ``loadcase``, pypownet/grid.py:65): a
    module with ONE function named after the file, returning the case dict; numbers at full precision (repr round-trips)."""
    fname = os.path.splitext(os.path.basename(path))[0]

    def rows(a):
        return ',\n'.join('        [' + ', '.join(repr(float(v)) for v in r) + ']' for r in np.asarray(a, dtype=np.float64))
    with open(path, 'w') as f:
        f.write('from numpy import array\n\n\ndef %s():\n    ppc = {"version": %r}\n\n' % (fname, str(ppc.get('version', '2'))))
        f.write('    ppc["baseMVA"] = %r\n\n' % float(ppc['baseMVA']))
        for k in ('bus', 'gen', 'branch'):
            f.write('    ppc[%r] = array([\n%s\n    ])\n\n' % (k, rows(ppc[k])))
        f.write('    return ppc\n')


def twin_id(sub_id):
    """External id of the artificial busbar of substation ``sub_id`` ('666' string prefix)."""
    return float(ARTIFICIAL_NODE_STARTING_STRING + str(int(sub_id)))


def id_to_substation(bus_id):
    """Inverse of the '666' encoding exactly as the reference does it: *every* occurrence of the
    substring is removed (``str.replace``; pypownet/grid.py:336-337, 377)."""
    return int(float(str(bus_id).replace(ARTIFICIAL_NODE_STARTING_STRING, '')))


class Case(object):
    """Static, index-based description of a doubled grid.

    Attributes (all numpy arrays, ``nS`` substations, ``nP`` productions, ``nL`` loads, ``nl`` lines):
      sub_ids[nS]            external substation ids (first half of bus[:,0]; strictly increasing)
      bus_gs/bs/basekv[2nS]  shunts / base voltage per bus *row* (row i+nS = twin of row i)
      vm0/va0[2nS]           case voltages (initial warm start; va0 in degrees as stored)
      gen_sub[nP]            substation index of each production (gen row order)
      gen_qmax/qmin[nP], gen_pg0/qg0/vg0[nP]
      load_sub[nL]           substation index of each load (ascending)
      load_pd0/qd0[nL]
      or_sub/ex_sub[nl], br_r/x/b/tap/shift[nl], br_status0[nl]
      slack_id               external id of the case's slack bus (pypownet/grid.py:74)
      slack_sub              its substation index
      n_elements[nS]         elements per substation; sub_elem_*: zipped (substation-major) permutation
                             (pypownet/grid.py:428-494)
    """

    def __init__(self, ppc):
        bus, gen, branch = ppc['bus'], ppc['gen'], ppc['branch']
        self.ppc = ppc
        self.baseMVA = float(ppc['baseMVA'])
        nrows = bus.shape[0]
        if nrows % 2:
            raise ValueError('reference grid must hold 2 bus rows per substation')
        nS = nrows // 2
        self.nS = nS
        self.sub_ids = bus[:nS, BUS_I].astype(np.int64)
        if not np.all(np.diff(self.sub_ids) > 0):
            raise ValueError('substation ids must be strictly increasing in the bus table')
        for i in range(nS):
            if bus[i + nS, BUS_I] != twin_id(self.sub_ids[i]):
                raise ValueError('bus row %d is not the 666-twin of row %d' % (i + nS, i))
        self.bus_gs = bus[:, GS].copy()
        self.bus_bs = bus[:, BS].copy()
        self.bus_basekv = bus[:, BASE_KV].copy()
        self.vm0 = bus[:, VM].copy()
        self.va0 = bus[:, VA].copy()
        self.bus_type0 = bus[:, BUS_TYPE].astype(np.int32)

        id2sub = {int(s): i for i, s in enumerate(self.sub_ids)}

        def to_sub_node(ids):
            subs, nodes = [], []
            for v in ids:
                iv = int(v)
                if iv in id2sub:
                    subs.append(id2sub[iv]); nodes.append(0)
                else:
                    s = id_to_substation(v)
                    subs.append(id2sub[s]); nodes.append(1)
            return np.asarray(subs, dtype=np.int32), np.asarray(nodes, dtype=np.int32)

        # productions
        self.nP = gen.shape[0]
        self.gen_sub, self.gen_node0 = to_sub_node(gen[:, GEN_BUS])
        if self.nP > 1 and not np.all(np.diff(self.gen_sub) > 0):
            raise ValueError('productions must be sorted by substation with at most one per substation')
        self.gen_pg0 = gen[:, PG].copy()
        self.gen_qg0 = gen[:, QG].copy()
        self.gen_vg0 = gen[:, VG].copy()
        self.gen_qmax = gen[:, QMAX].copy()
        self.gen_qmin = gen[:, QMIN].copy()
        self.gen_status0 = (gen[:, GEN_STATUS] > 0).astype(np.int32)

        # loads: rows with Pd != 0 or Qd != 0 (pypownet/grid.py:77)
        are_loads = np.logical_or(bus[:, PD] != 0, bus[:, QD] != 0)
        rows = np.where(are_loads)[0]
        self.nL = len(rows)
        self.load_sub = (rows % nS).astype(np.int32)
        self.load_node0 = (rows // nS).astype(np.int32)
        if self.nL > 1 and not np.all(np.diff(self.load_sub) > 0):
            raise ValueError('at most one load per substation is supported (reference assumption)')
        self.load_pd0 = bus[rows, PD].copy()
        self.load_qd0 = bus[rows, QD].copy()

        # lines
        self.nl = branch.shape[0]
        self.or_sub, self.or_node0 = to_sub_node(branch[:, F_BUS])
        self.ex_sub, self.ex_node0 = to_sub_node(branch[:, T_BUS])
        self.br_r = branch[:, BR_R].copy()
        self.br_x = branch[:, BR_X].copy()
        self.br_b = branch[:, BR_B].copy()
        self.br_tap = branch[:, TAP].copy()
        self.br_shift = branch[:, SHIFT].copy()
        self.br_status0 = (branch[:, BR_STATUS] != 0).astype(np.int32)

        # slack (pypownet/grid.py:74): id of the first row whose type is 3
        w = np.where(bus[:, BUS_TYPE] == 3)[0]
        if len(w) == 0:
            raise ValueError('case has no slack bus')
        self.slack_id = int(bus[w[0], BUS_I])
        self.slack_row = int(w[0])
        self.slack_sub = int(w[0] % nS)

        self._build_zipped_mapping()

    # ------------------------------------------------------------------------------------------
    def _build_zipped_mapping(self):
        """Substation-major ("zipped") ordering of the topology vector
        [prods | loads | lines_or | lines_ex] (pypownet/grid.py:428-494, 598-624): per substation
        [prod?, load?, origins..., extremities...]."""
        nP, nL, nl = self.nP, self.nL, self.nl
        mapping, n_elements = [], []
        for s in range(self.nS):
            m = []
            m.extend(np.where(self.gen_sub == s)[0][:1].tolist())
            m.extend((np.where(self.load_sub == s)[0][:1] + nP).tolist())
            m.extend((np.where(self.or_sub == s)[0] + nP + nL).tolist())
            m.extend((np.where(self.ex_sub == s)[0] + nP + nL + nl).tolist())
            mapping.append(m)
            n_elements.append(len(m))
        self.mapping_array = mapping
        self.n_elements = np.asarray(n_elements, dtype=np.int32)
        self.zip_perm = np.asarray([c for m in mapping for c in m], dtype=np.int32)
        inv = np.empty_like(self.zip_perm)
        inv[self.zip_perm] = np.arange(len(self.zip_perm), dtype=np.int32)
        self.unzip_perm = inv
        # substation of every element of the unzipped topology vector
        self.elem_sub = np.concatenate([self.gen_sub, self.load_sub, self.or_sub, self.ex_sub]).astype(np.int32)
        self.n_topo = nP + nL + 2 * nl

    @property
    def action_length(self):
        return self.nP + self.nL + 3 * self.nl

    @property
    def observation_length(self):
        # environment.py:451-466, 511-517, 583-595: 9 nL + 9 nP + 18 nl + 2 nS + 6
        return 9 * self.nL + 9 * self.nP + 18 * self.nl + 2 * self.nS + 6

    # external ids as the reference would print them -------------------------------------------
    def bus_row_id(self, sub, node):
        return float(self.sub_ids[sub]) if node == 0 else twin_id(self.sub_ids[sub])

    @classmethod
    def from_file(cls, path):
        return cls(load_case_file(path))
