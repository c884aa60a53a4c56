"""Chronic (time-series) input format of pypownet, re-implemented.

Behaviour follows the reference reader (pypownet/chronic.py:124-257): 13 ';'-separated CSV files with one
ignored header row, every numeric file parsed as float32, the four ``*_planned`` series shifted up by
one row with the last row duplicated, the number of timesteps given by the *shortest* of the zipped
series, ids unique.  In addition to CSV folders a compact ``<name>.npz`` cache of the very same arrays
is accepted (binary chronic cache) so that fixtures stay small.
"""
import os
from datetime import datetime

import numpy as np

_NUMERIC = {
    'loads_p': '_N_loads_p.csv', 'loads_q': '_N_loads_q.csv',
    'prods_p': '_N_prods_p.csv', 'prods_v': '_N_prods_v.csv',
    'loads_p_planned': '_N_loads_p_planned.csv', 'loads_q_planned': '_N_loads_q_planned.csv',
    'prods_p_planned': '_N_prods_p_planned.csv', 'prods_v_planned': '_N_prods_v_planned.csv',
    'ids': '_N_simu_ids.csv', 'imaps': '_N_imaps.csv',
    'maintenance': 'maintenance.csv', 'hazards': 'hazards.csv',
}
_DATETIMES = '_N_datetimes.csv'


def _csv(path):
    # same call as the reference (chronic.py:175): float32, ';', header skipped
    return np.genfromtxt(path, dtype=np.float32, delimiter=';', skip_header=True)


class TimestepEntries(object):
    """One row of a chronic (pypownet/chronic.py:12-67)."""

    def __init__(self, timestep_id, loads_p, loads_q, prods_p, prods_v, maintenance, hazards, date,
                 planned_loads_p=None, planned_loads_q=None, planned_prods_p=None, planned_prods_v=None):
        self.id = timestep_id
        self.prods_p, self.prods_v, self.loads_p, self.loads_q = prods_p, prods_v, loads_p, loads_q
        self.planned_prods_p, self.planned_prods_v = planned_prods_p, planned_prods_v
        self.planned_loads_p, self.planned_loads_q = planned_loads_p, planned_loads_q
        self.maintenance, self.hazards = maintenance, hazards
        self.datetime = datetime.strptime(date.lower(), '%Y-%b-%d;%H:%M')

    def get_prods_p(self): return self.prods_p
    def get_prods_v(self): return self.prods_v
    def get_loads_p(self): return self.loads_p
    def get_loads_q(self): return self.loads_q
    def get_planned_prods_p(self): return self.planned_prods_p
    def get_planned_prods_v(self): return self.planned_prods_v
    def get_planned_loads_p(self): return self.planned_loads_p
    def get_planned_loads_q(self): return self.planned_loads_q
    def get_id(self): return self.id
    def get_maintenance(self): return self.maintenance
    def get_hazards(self): return self.hazards
    def get_datetime(self): return self.datetime


_MONTHS = {m: k + 1 for k, m in enumerate(('jan', 'feb', 'mar', 'apr', 'may', 'jun', 'jul', 'aug', 'sep', 'oct', 'nov', 'dec'))}
_CACHE = {}


def load_chronic(source, with_previsions=True):
    """``Chronic(source)`` through a cache keyed by path and modification time: RunEnv.reset() re-creates the Game like the
    reference does (environment.py:814-821) and would otherwise re-parse every chronic of the level each time.  The
    arrays of a cached chronic are shared; nothing in the package writes to them."""
    key = (os.path.abspath(source), os.path.getmtime(source), bool(with_previsions))
    c = _CACHE.get(key)
    if c is None:
        if len(_CACHE) > 256:
            _CACHE.clear()
        c = _CACHE[key] = Chronic(source, with_previsions)
    return c


class Chronic(object):
    """A whole chronic held as dense ``[T x n]`` float32 arrays (the layout uploaded to the GPU)."""

    def __init__(self, source, with_previsions=True):
        if not os.path.exists(source):
            raise ValueError('Source folder %s does not exist' % source)
        self.source_folder = source
        self.with_previsions = with_previsions
        if os.path.isdir(source):
            self.name = os.path.basename(os.path.normpath(source))
            data, datetimes = self._read_folder(source)
        else:
            self.name = os.path.splitext(os.path.basename(source))[0]
            data, datetimes = self._read_npz(source)
        self._import(data, datetimes)

    # -- readers ---------------------------------------------------------------------------------
    @staticmethod
    def _read_folder(folder):
        present = [f for f in os.listdir(folder) if os.path.isfile(os.path.join(folder, f)) and f.endswith('.csv')]
        for fname in list(_NUMERIC.values()) + [_DATETIMES]:
            if fname not in present:
                raise FileExistsError('File %s does not exist but is mandatory' % fname)
        data = {k: _csv(os.path.join(folder, f)) for k, f in _NUMERIC.items()}
        with open(os.path.join(folder, _DATETIMES), 'r') as f:
            datetimes = f.read().splitlines()[1:]
        return data, datetimes

    @staticmethod
    def _read_npz(path):
        z = np.load(path, allow_pickle=False)
        data = {k: z[k].astype(np.float32) for k in _NUMERIC}
        datetimes = [str(s) for s in z['datetimes'].tolist()]
        return data, datetimes

    def to_npz(self, path, max_rows=None):
        """Write the raw (un-shifted) series; ``max_rows`` truncates every per-timestep file."""
        raw = self._raw
        out = {}
        for k in _NUMERIC:
            a = raw[k]
            if k != 'imaps' and max_rows is not None:
                a = a[:max_rows]
            out[k] = a
        dts = self._raw_datetimes if max_rows is None else self._raw_datetimes[:max_rows]
        np.savez_compressed(path, datetimes=np.asarray(dts), **out)

    # -- reference semantics ---------------------------------------------------------------------
    def _import(self, data, datetimes):
        self._raw = {k: np.array(v, copy=True) for k, v in data.items()}
        self._raw_datetimes = list(datetimes)
        two_d = lambda a: np.atleast_2d(a)
        self.prods_p = two_d(data['prods_p']); self.prods_v = two_d(data['prods_v'])
        self.loads_p = two_d(data['loads_p']); self.loads_q = two_d(data['loads_q'])
        self.prods_p_planned = two_d(np.array(data['prods_p_planned'], copy=True))
        self.prods_v_planned = two_d(np.array(data['prods_v_planned'], copy=True))
        self.loads_p_planned = two_d(np.array(data['loads_p_planned'], copy=True))
        self.loads_q_planned = two_d(np.array(data['loads_q_planned'], copy=True))
        self.imaps = np.atleast_1d(data['imaps']).tolist()
        # slip planned series by one (chronic.py:202-205): row t carries the forecast for t+1
        for a in (self.prods_p_planned, self.prods_v_planned, self.loads_p_planned, self.loads_q_planned):
            a[:-1] = a[1:]
        self.maintenance = two_d(data['maintenance'])
        self.hazards = two_d(data['hazards'])
        self.timestep_ids = np.atleast_1d(data['ids']).astype(np.int32).tolist()
        self.datetimes = list(datetimes)
        assert len(np.unique(self.timestep_ids)) == len(self.timestep_ids), 'There are timesteps with the same id'
        # zip() truncation to the shortest series (chronic.py:225-229)
        self.n_timesteps = min(len(self.timestep_ids), len(self.loads_p), len(self.loads_q), len(self.prods_p),
                               len(self.prods_v), len(self.loads_p_planned), len(self.loads_q_planned),
                               len(self.prods_p_planned), len(self.prods_v_planned), len(self.maintenance),
                               len(self.hazards), len(self.datetimes))
        self._entries = {}
        self._parsed_dates = None

    # -- accessors (reference API) ---------------------------------------------------------------
    def get_timestep_ids(self):
        return self.timestep_ids

    def get_imaps(self):
        return self.imaps

    def _row_of(self, timestep_id):
        if timestep_id not in self.timestep_ids:
            raise ValueError('Could not find TimestepInjections with id', timestep_id)
        return self.timestep_ids.index(timestep_id)

    def get_timestep_entries(self, timestep_id):
        r = self._row_of(timestep_id)
        if r >= self.n_timesteps:
            raise IndexError('list index out of range')
        if r not in self._entries:
            self._entries[r] = TimestepEntries(
                self.timestep_ids[r], self.loads_p[r], self.loads_q[r], self.prods_p[r], self.prods_v[r],
                self.maintenance[r], self.hazards[r], self.datetimes[r], self.loads_p_planned[r],
                self.loads_q_planned[r], self.prods_p_planned[r], self.prods_v_planned[r])
        return self._entries[r]

    def get_planned_maintenance(self, timestep_id, horizon):
        """Timesteps before the next maintenance of each line within the horizon (chronic.py:239-246)."""
        b = self._row_of(timestep_id)
        m = self.maintenance[b:min(b + horizon, self.n_timesteps)]
        return (m != 0).argmax(axis=0)

    def get_timestep_duration(self):
        d0 = self.get_timestep_entries(self.timestep_ids[0]).get_datetime()
        d1 = self.get_timestep_entries(self.timestep_ids[1]).get_datetime()
        return (d1 - d0).total_seconds()

    def date_fields(self):
        """[T x 6] int32 (year, month, day, hour, minute, second) for the device observation gather."""
        if self._parsed_dates is None:
            out = np.zeros((self.n_timesteps, 6), dtype=np.int32)
            for r in range(self.n_timesteps):
                s = self.datetimes[r].lower()
                try:        # 'YYYY-mon-DD;HH:MM' split by hand (strptime costs 6 us per row, every RunEnv.reset re-reads them)
                    day, clock = s.split(';')
                    y, mon, dd = day.split('-')
                    hh, mm = clock.split(':')
                    out[r] = (int(y), _MONTHS[mon], int(dd), int(hh), int(mm), 0)
                except (ValueError, KeyError):
                    d = datetime.strptime(s, '%Y-%b-%d;%H:%M')
                    out[r] = (d.year, d.month, d.day, d.hour, d.minute, d.second)
            self._parsed_dates = out
        return self._parsed_dates


class ChronicLooper(object):
    """Chronic folder iteration (pypownet/chronic.py:260-295): sorted by name, natural/random/fixed."""

    def __init__(self, chronics_folder, game_level, start_id, looping_mode):
        self.chronics_folder = os.path.abspath(chronics_folder)
        if not os.path.exists(self.chronics_folder):
            raise FileNotFoundError('Chronic folder %s does not exist' % self.chronics_folder)
        if looping_mode not in ['natural', 'random', 'fixed']:
            raise ValueError('Either "natural" mode (loops in the order of chronics ids), "random" (loops randomly) or'
                             '"fixed" (plays the same chronic)')
        self.looping_mode = looping_mode
        entries = []
        for d in os.listdir(self.chronics_folder):
            p = os.path.join(self.chronics_folder, d)
            if os.path.isdir(p) or d.endswith('.npz'):
                entries.append(p)
        self.chronics = sorted(entries)
        self.next_chronic_id = start_id if self.looping_mode != 'random' else np.random.choice(len(self.chronics))
        self.current_chronic_name = None

    def get_next_chronic_folder(self):
        res = self.chronics[self.next_chronic_id]
        self.current_chronic_name = os.path.splitext(os.path.basename(res))[0]
        if self.looping_mode == 'natural':
            self.next_chronic_id = (self.next_chronic_id + 1) % len(self.chronics)
        elif self.looping_mode == 'random':
            self.next_chronic_id = np.random.choice(len(self.chronics))
        return res

    def get_current_chronic_name(self):
        return self.current_chronic_name
