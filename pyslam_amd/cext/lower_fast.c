/* lower_fast.c -- the walk over a Problem's reprojection blocks in C (CPython API; host code, no GPU).
 *
 * pyslam_amd/lowering.py: lower() turns the reference's object graph (one Python object per residual block, reference
 * pyslam/problem.py:43-108) into flat tables once per solve.  For a bundle adjustment that is 5 x 10^5 blocks of ONE kind
 * (ReprojectionResidual: pose key, landmark key, observation, shared camera / stiffness / loss), and the interpreter's
 * ~0.6 us per block was what a cold Problem.solve() spent most of its wall clock in.  walk() takes a run of consecutive
 * 'reproj' blocks straight into the column arrays; anything it does not recognise -- another kind of block, an unusual
 * key, an observation that is not three contiguous doubles -- ends the run and is handed back to the Python loop, which
 * treats that one block exactly as before (and raises what it raised before).  Same tables either way
 * (tests/test_lowering_fast.py).
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>
#include <pthread.h>
#include <stdlib.h>
/* Round 6: block.obs through the numpy C API (PyArray_DATA after a type / shape / layout check) instead of the buffer protocol,
 * whose numpy implementation builds and caches a format string per array -- one allocation per block on the first lowering
 * (500 000 blocks: ~35 of the walk's 80 ms); -DPS_LOWER_NO_NUMPY falls back to the buffer protocol (no numpy headers). */
#ifndef PS_LOWER_NO_NUMPY
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>
#endif

static PyObject *s_KIND, *s_camera, *s_stiffness, *s_obs, *s_CAMERA_ID;

typedef struct { PyObject *cam, *stiff, *loss; long g; } GroupSlot;

/* An instance attribute by its interned name: straight from the instance dictionary when the TYPE says that is what getattr
 * would return (generic getattr, no descriptor of that name anywhere in the MRO -- checked once per type by plain_attrs()), the
 * general protocol otherwise.  -> NEW reference or NULL (error cleared by the caller). */
static PyObject* inst_attr(PyObject* o, PyObject* name, int plain) {
    if (plain) {
        PyObject** dp = _PyObject_GetDictPtr(o);
        if (dp && *dp) {
            PyObject* v = PyDict_GetItemWithError(*dp, name);      /* borrowed */
            if (v) { Py_INCREF(v); return v; }
            if (PyErr_Occurred()) return NULL;
        }
    }
    return PyObject_GetAttr(o, name);
}
static int plain_attrs(PyTypeObject* tp) {
    return tp->tp_getattro == PyObject_GenericGetAttr && !_PyType_Lookup(tp, s_camera) && !_PyType_Lookup(tp, s_stiffness) &&
           !_PyType_Lookup(tp, s_obs);
}

/* ---- the walk on several threads (round 6) ----------------------------------------------------------------------------------
 * The walk is bound by the latency of the interpreter's object graph (~75 ns per block over five levels of pointers), not by work:
 * it scales with threads.  The calling thread HOLDS THE GIL and waits in pthread_join, so no Python code runs and nothing is mutated
 * while the workers run; the workers only READ -- borrowed references, no reference count is touched, no Python API that can
 * allocate, raise or run user code is called (exact str keys with cached hashes through _PyDict_GetItem_KnownHash, exact ndarrays
 * through the array struct, exact small ints) -- and write rows of the caller's column arrays at positions fixed by the block index.
 * Anything irregular (another type, a key without a cached hash, a group not yet known, an observation that is not three doubles)
 * ends a worker's chunk; the first such block over all chunks is where the caller continues serially, exactly as before.
 * PYSLAM_AMD_LOWER_THREADS=1 switches it off (lowering.py passes the thread count). */
typedef struct {
    PyObject **blocks, **keys, **losses;
    Py_ssize_t lo, hi, stop;                 /* chunk [lo, hi); stop = first block NOT taken */
    Py_ssize_t row0, i_base;                 /* block i goes to row row0 + (i - i_base) */
    PyTypeObject* reproj_tp;
    PyObject *pose_ix, *point_ix;
    const GroupSlot* slots; int nslots;
    int32_t *o_pose, *o_pt, *o_g; double* o_uvd;
    Py_hash_t h_camera, h_stiffness, h_obs;
} WalkChunk;

static inline Py_hash_t cached_str_hash(PyObject* k) {
    return PyUnicode_CheckExact(k) ? ((PyASCIIObject*)k)->hash : -1;
}

static void* walk_chunk(void* arg) {
    WalkChunk* c = (WalkChunk*)arg;
    Py_ssize_t i = c->lo;
    for (; i < c->hi; ++i) {
        PyObject* block = c->blocks[i];
        if (Py_TYPE(block) != c->reproj_tp) break;
        PyObject* ks = c->keys[i];
        PyObject *k0, *k1;
        if (PyList_CheckExact(ks) && PyList_GET_SIZE(ks) == 2) { k0 = PyList_GET_ITEM(ks, 0); k1 = PyList_GET_ITEM(ks, 1); }
        else if (PyTuple_CheckExact(ks) && PyTuple_GET_SIZE(ks) == 2) { k0 = PyTuple_GET_ITEM(ks, 0); k1 = PyTuple_GET_ITEM(ks, 1); }
        else break;
        const Py_hash_t h0 = cached_str_hash(k0), h1 = cached_str_hash(k1);
        if (h0 == -1 || h1 == -1) break;
        PyObject* pi = _PyDict_GetItem_KnownHash(c->pose_ix, k0, h0);
        PyObject* qi = pi ? _PyDict_GetItem_KnownHash(c->point_ix, k1, h1) : NULL;
        if (!pi || !qi || !PyLong_CheckExact(pi) || !PyLong_CheckExact(qi)) break;
        PyObject** dp = _PyObject_GetDictPtr(block);
        if (!dp || !*dp) break;
        PyObject* cam = _PyDict_GetItem_KnownHash(*dp, s_camera, c->h_camera);
        PyObject* stiff = cam ? _PyDict_GetItem_KnownHash(*dp, s_stiffness, c->h_stiffness) : NULL;
        PyObject* obs = stiff ? _PyDict_GetItem_KnownHash(*dp, s_obs, c->h_obs) : NULL;
        if (!obs) break;
        PyObject* loss = c->losses[i];
        long g = -1;
        for (int q = 0; q < c->nslots; ++q)
            if (c->slots[q].cam == cam && c->slots[q].stiff == stiff && c->slots[q].loss == loss) { g = c->slots[q].g; break; }
        if (g < 0) break;
        const Py_ssize_t row = c->row0 + (i - c->i_base);
#ifndef PS_LOWER_NO_NUMPY
        if (!PyArray_CheckExact(obs)) break;
        PyArrayObject* a = (PyArrayObject*)obs;
        if (!(PyArray_TYPE(a) == NPY_DOUBLE && PyArray_ISNOTSWAPPED(a) && PyArray_ISALIGNED(a) && PyArray_SIZE(a) == 3 && PyArray_IS_C_CONTIGUOUS(a))) break;
        memcpy(c->o_uvd + 3 * row, PyArray_DATA(a), 24);
#else
        break;
#endif
        /* exact ints of at most one 30-bit digit (indices into tables of < 2^30 rows): read from the object, no API call */
        const Py_ssize_t sp = Py_SIZE(pi), sq = Py_SIZE(qi);
        if (sp < 0 || sp > 1 || sq < 0 || sq > 1) break;
        c->o_pose[row] = sp ? (int32_t)((PyLongObject*)pi)->ob_digit[0] : 0;
        c->o_pt[row] = sq ? (int32_t)((PyLongObject*)qi)->ob_digit[0] : 0;
        c->o_g[row] = (int32_t)g;
    }
    c->stop = i;
    return NULL;
}

/* walk(blocks, keys, losses, i0, param_dict, pose_ix, point_ix, group_of, o_pose, o_pt, o_uvd, o_g, count)
 * -> (i, count): blocks i0 .. i-1 were taken; block i (if i < len) is for the caller */
static PyObject* walk(PyObject* self, PyObject* args) {
    PyObject *blocks, *keys, *losses, *param_dict, *pose_ix, *point_ix, *group_of, *a_pose, *a_pt, *a_uvd, *a_g;
    Py_ssize_t i0, count;
    int nthreads = 1;
    if (!PyArg_ParseTuple(args, "OOOnOOOOOOOOn|i", &blocks, &keys, &losses, &i0, &param_dict, &pose_ix, &point_ix, &group_of,
                          &a_pose, &a_pt, &a_uvd, &a_g, &count, &nthreads)) return NULL;
    if (!PyList_Check(blocks) || !PyList_Check(keys) || !PyList_Check(losses) || !PyDict_Check(param_dict) ||
        !PyDict_Check(pose_ix) || !PyDict_Check(point_ix))
        return Py_BuildValue("nn", i0, count);
    Py_buffer b_pose, b_pt, b_uvd, b_g;
    if (PyObject_GetBuffer(a_pose, &b_pose, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS)) return NULL;
    if (PyObject_GetBuffer(a_pt, &b_pt, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS)) { PyBuffer_Release(&b_pose); return NULL; }
    if (PyObject_GetBuffer(a_uvd, &b_uvd, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS)) { PyBuffer_Release(&b_pose); PyBuffer_Release(&b_pt); return NULL; }
    if (PyObject_GetBuffer(a_g, &b_g, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS)) { PyBuffer_Release(&b_pose); PyBuffer_Release(&b_pt); PyBuffer_Release(&b_uvd); return NULL; }
    int32_t *o_pose = (int32_t*)b_pose.buf, *o_pt = (int32_t*)b_pt.buf, *o_g = (int32_t*)b_g.buf;
    double* o_uvd = (double*)b_uvd.buf;
    const Py_ssize_t cap = b_pose.len / 4;
    Py_ssize_t n = PyList_GET_SIZE(blocks);
    if (PyList_GET_SIZE(keys) < n) n = PyList_GET_SIZE(keys);
    if (PyList_GET_SIZE(losses) < n) n = PyList_GET_SIZE(losses);
    int failed = 0;
    if (b_pt.len / 4 < cap || b_g.len / 4 < cap || b_uvd.len / 24 < cap) { PyErr_SetString(PyExc_ValueError, "lower_fast.walk: column arrays of different capacity"); failed = 1; }

    PyTypeObject *reproj_tp = NULL, *other_tp = NULL;
    int plain = 0;
    PyObject* cam_ok = NULL;                 /* the camera whose CAMERA_ID was last found to be 0 or 1 (borrowed; compared by address only) */
    GroupSlot slots[4];
    int nslots = 0;
    Py_ssize_t i = i0;
    for (; i < n && !failed; ++i) {
        if (count >= cap) break;
        /* (round 6) a long run, the type known to have plain attributes, the group table warm: the rest on several threads */
        if (nthreads > 1 && plain && reproj_tp && nslots > 0 && i - i0 >= 256 && n - i >= 65536 && cap - count >= n - i) {
            enum { MAXT = 16 };
            const int T = nthreads > MAXT ? MAXT : nthreads;
            WalkChunk ch[MAXT];
            pthread_t th[MAXT];
            const Py_ssize_t span = n - i;
            const Py_hash_t hc = cached_str_hash(s_camera), hs = cached_str_hash(s_stiffness), ho = cached_str_hash(s_obs);
            int started = 0;
            if (hc != -1 && hs != -1 && ho != -1) {
                for (int t = 0; t < T; ++t) {
                    WalkChunk* c = &ch[t];
                    c->blocks = ((PyListObject*)blocks)->ob_item; c->keys = ((PyListObject*)keys)->ob_item; c->losses = ((PyListObject*)losses)->ob_item;
                    c->lo = i + span * t / T; c->hi = i + span * (t + 1) / T; c->stop = c->lo;
                    c->row0 = count; c->i_base = i; c->reproj_tp = reproj_tp; c->pose_ix = pose_ix; c->point_ix = point_ix;
                    c->slots = slots; c->nslots = nslots; c->o_pose = o_pose; c->o_pt = o_pt; c->o_g = o_g; c->o_uvd = o_uvd;
                    c->h_camera = hc; c->h_stiffness = hs; c->h_obs = ho;
                }
                for (int t = 1; t < T; ++t) { if (pthread_create(&th[t], NULL, walk_chunk, &ch[t])) break; started = t; }
                walk_chunk(&ch[0]);
                for (int t = 1; t <= started; ++t) pthread_join(th[t], NULL);
                /* blocks are taken up to the first one some chunk did not take (a chunk whose thread could not be started took none) */
                Py_ssize_t stop = ch[0].stop;
                if (stop == ch[0].hi) for (int t = 1; t < T; ++t) { if (t > started) break; stop = ch[t].stop; if (stop != ch[t].hi) break; }
                count += stop - i;
                i = stop;
                nthreads = 1;                    /* (whatever is left: serially, from the block that ended the run) */
                if (i >= n) break;
            } else nthreads = 1;
        }
        /* the walk is bound by the latency of the interpreter's object graph (block -> its dictionary -> the values -> obs -> data):
         * ask for the next blocks' first levels while this one is taken */
        if (i + 12 < n) { __builtin_prefetch(PyList_GET_ITEM(blocks, i + 12)); __builtin_prefetch(PyList_GET_ITEM(keys, i + 12)); }
        if (plain && i + 6 < n) {
            PyObject* b6 = PyList_GET_ITEM(blocks, i + 6);
            if (Py_TYPE(b6) == reproj_tp) { PyObject** dp6 = _PyObject_GetDictPtr(b6); if (dp6 && *dp6) __builtin_prefetch(*dp6); }
        }
        if (plain && i + 3 < n) {
            PyObject* b3 = PyList_GET_ITEM(blocks, i + 3);
            if (Py_TYPE(b3) == reproj_tp) {
                PyObject** dp3 = _PyObject_GetDictPtr(b3);
                if (dp3 && *dp3 && ((PyDictObject*)*dp3)->ma_values) __builtin_prefetch(((PyDictObject*)*dp3)->ma_values);
            }
        }
        PyObject* block = PyList_GET_ITEM(blocks, i);
        PyTypeObject* tp = Py_TYPE(block);
        if (tp != reproj_tp) {
            if (tp == other_tp) break;
            PyObject* kind = PyObject_GetAttr(block, s_KIND);
            if (!kind) { PyErr_Clear(); break; }
            const int is_reproj = PyUnicode_Check(kind) && PyUnicode_CompareWithASCIIString(kind, "reproj") == 0;
            Py_DECREF(kind);
            if (!is_reproj) { other_tp = tp; break; }
            reproj_tp = tp;
            plain = plain_attrs(tp);
        }
        PyObject* ks = PyList_GET_ITEM(keys, i);
        PyObject *k0, *k1;
        if (PyList_Check(ks) && PyList_GET_SIZE(ks) == 2) { k0 = PyList_GET_ITEM(ks, 0); k1 = PyList_GET_ITEM(ks, 1); }
        else if (PyTuple_Check(ks) && PyTuple_GET_SIZE(ks) == 2) { k0 = PyTuple_GET_ITEM(ks, 0); k1 = PyTuple_GET_ITEM(ks, 1); }
        else break;
        /* (pose_ix / point_ix hold exactly the keys of param_dict, split by kind -- lowering.py: classify -- so a key found there IS
         *  a parameter: no second look into param_dict itself; a key found in neither ends the run and the Python loop says why) */
        PyObject* pi = PyDict_GetItemWithError(pose_ix, k0);       /* borrowed */
        PyObject* qi = pi ? PyDict_GetItemWithError(point_ix, k1) : NULL;
        if (!pi || !qi) { PyErr_Clear(); break; }
        PyObject* cam = inst_attr(block, s_camera, plain);
        if (!cam) { PyErr_Clear(); break; }
        if (cam != cam_ok) {
            PyObject* cid = PyObject_GetAttr(cam, s_CAMERA_ID);
            long v = -1;
            if (cid) { v = PyLong_Check(cid) ? PyLong_AsLong(cid) : -1; Py_DECREF(cid); }
            PyErr_Clear();
            if (v != 0 && v != 1) { Py_DECREF(cam); break; }
            cam_ok = cam;
        }
        PyObject* stiff = inst_attr(block, s_stiffness, plain);
        if (!stiff) { PyErr_Clear(); Py_DECREF(cam); break; }
        PyObject* loss = PyList_GET_ITEM(losses, i);
        long g = -1;
        for (int q = 0; q < nslots; ++q)
            if (slots[q].cam == cam && slots[q].stiff == stiff && slots[q].loss == loss) { g = slots[q].g; break; }
        if (g < 0) {
            /* the Python side owns the group table (and its checks): identity-keyed, so the three objects stay alive there */
            PyObject* r = PyObject_CallFunctionObjArgs(group_of, cam, block, loss, NULL);
            if (!r) { Py_DECREF(cam); Py_DECREF(stiff); failed = 1; break; }
            g = PyLong_AsLong(r);
            Py_DECREF(r);
            if (g < 0) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "lower_fast.walk: bad group index"); Py_DECREF(cam); Py_DECREF(stiff); failed = 1; break; }
            GroupSlot* sl = &slots[nslots < 4 ? nslots++ : (int)(i & 3)];
            sl->cam = cam; sl->stiff = stiff; sl->loss = loss; sl->g = g;
        }
        Py_DECREF(cam); Py_DECREF(stiff);          /* (still referenced by the block; the slots compare addresses only) */
        PyObject* obs = inst_attr(block, s_obs, plain);
        if (!obs) { PyErr_Clear(); break; }
        int good = 0;
#ifndef PS_LOWER_NO_NUMPY
        if (PyArray_CheckExact(obs)) {           /* three native doubles in a row: straight from the array's data */
            PyArrayObject* a = (PyArrayObject*)obs;
            good = PyArray_TYPE(a) == NPY_DOUBLE && PyArray_ISNOTSWAPPED(a) && PyArray_ISALIGNED(a) && PyArray_SIZE(a) == 3 &&
                   PyArray_IS_C_CONTIGUOUS(a);
            if (good) memcpy(o_uvd + 3 * count, PyArray_DATA(a), 24);
        } else
#endif
        {
            Py_buffer ov;
            if (PyObject_GetBuffer(obs, &ov, PyBUF_FORMAT | PyBUF_C_CONTIGUOUS)) { PyErr_Clear(); Py_DECREF(obs); break; }
            good = ov.itemsize == 8 && ov.len == 24 && ov.format && (strcmp(ov.format, "d") == 0 || strcmp(ov.format, "=d") == 0 || strcmp(ov.format, "<d") == 0);
            if (good) memcpy(o_uvd + 3 * count, ov.buf, 24);
            PyBuffer_Release(&ov);
        }
        Py_DECREF(obs);
        if (!good) break;
        const long pv = PyLong_AsLong(pi), qv = PyLong_AsLong(qi);
        if ((pv == -1 || qv == -1) && PyErr_Occurred()) { PyErr_Clear(); break; }
        o_pose[count] = (int32_t)pv; o_pt[count] = (int32_t)qv; o_g[count] = (int32_t)g;
        ++count;
    }
    PyBuffer_Release(&b_pose); PyBuffer_Release(&b_pt); PyBuffer_Release(&b_uvd); PyBuffer_Release(&b_g);
    if (failed) return NULL;
    return Py_BuildValue("nn", i, count);
}

/* classify(param_dict, pose_types: tuple, ndarray_type) -> (pose_keys, point_keys, offending key or None): the dictionary's keys in
 * its own order, split into poses (instances of pose_types) and landmarks (ndarrays of shape (3,)); stops at the first value that
 * is neither */
static PyObject* classify(PyObject* self, PyObject* args) {
    PyObject *param_dict, *pose_types, *nd_type;
    if (!PyArg_ParseTuple(args, "O!OO", &PyDict_Type, &param_dict, &pose_types, &nd_type)) return NULL;
    PyObject *poses = PyList_New(0), *points = PyList_New(0), *badkey = Py_None;
    /* (round 6) the index dictionaries key -> position of the two lists, filled in the same pass (they were two dictionary
     * comprehensions over 50 000 keys on the Python side: 4 ms of a C3 lowering) */
    PyObject *pose_ix = PyDict_New(), *point_ix = PyDict_New();
    if (!poses || !points || !pose_ix || !point_ix) { Py_XDECREF(poses); Py_XDECREF(points); Py_XDECREF(pose_ix); Py_XDECREF(point_ix); return NULL; }
    Py_ssize_t pos = 0;
    PyObject *key, *val;
    PyTypeObject* last_pose_tp = NULL;
    while (PyDict_Next(param_dict, &pos, &key, &val)) {
        int is_pose = Py_TYPE(val) == last_pose_tp;
        if (!is_pose && (PyObject*)Py_TYPE(val) != nd_type) {
            is_pose = PyObject_IsInstance(val, pose_types);
            if (is_pose < 0) { Py_DECREF(poses); Py_DECREF(points); Py_DECREF(pose_ix); Py_DECREF(point_ix); return NULL; }
            if (is_pose) last_pose_tp = Py_TYPE(val);
        }
        if (is_pose) {
            PyObject* ix = PyLong_FromSsize_t(PyList_GET_SIZE(poses));
            if (!ix || PyDict_SetItem(pose_ix, key, ix) || PyList_Append(poses, key)) { Py_XDECREF(ix); Py_DECREF(poses); Py_DECREF(points); Py_DECREF(pose_ix); Py_DECREF(point_ix); return NULL; }
            Py_DECREF(ix);
            continue;
        }
        int ok = 0;
        const int is_nd = (PyObject*)Py_TYPE(val) == nd_type ? 1 : PyObject_IsInstance(val, nd_type);
        if (is_nd < 0) { Py_DECREF(poses); Py_DECREF(points); Py_DECREF(pose_ix); Py_DECREF(point_ix); return NULL; }
        if (is_nd) {
#ifndef PS_LOWER_NO_NUMPY
            if (PyArray_Check(val)) ok = PyArray_NDIM((PyArrayObject*)val) == 1 && PyArray_DIM((PyArrayObject*)val, 0) == 3;
            else
#endif
            {
                Py_buffer v;
                if (PyObject_GetBuffer(val, &v, PyBUF_STRIDES) == 0) { ok = v.ndim == 1 && v.shape[0] == 3; PyBuffer_Release(&v); }
                else PyErr_Clear();
            }
        }
        if (!ok) { badkey = key; break; }
        {
            PyObject* ix = PyLong_FromSsize_t(PyList_GET_SIZE(points));
            if (!ix || PyDict_SetItem(point_ix, key, ix) || PyList_Append(points, key)) { Py_XDECREF(ix); Py_DECREF(poses); Py_DECREF(points); Py_DECREF(pose_ix); Py_DECREF(point_ix); return NULL; }
            Py_DECREF(ix);
        }
    }
    return Py_BuildValue("NNONN", poses, points, badkey, pose_ix, point_ix);
}

static int is_f64x3(const Py_buffer* v) {
    return v->itemsize == 8 && v->len == 24 && v->format && (strcmp(v->format, "d") == 0 || strcmp(v->format, "=d") == 0 || strcmp(v->format, "<d") == 0);
}

/* gather3(param_dict, keys: list, dst: (n, 3) float64) -> list of the indices NOT copied (value is not three contiguous doubles) */
static PyObject* gather3(PyObject* self, PyObject* args) {
    PyObject *param_dict, *keys, *dst;
    if (!PyArg_ParseTuple(args, "O!O!O", &PyDict_Type, &param_dict, &PyList_Type, &keys, &dst)) return NULL;
    Py_buffer b;
    if (PyObject_GetBuffer(dst, &b, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS)) return NULL;
    const Py_ssize_t n = PyList_GET_SIZE(keys);
    PyObject* rest = PyList_New(0);
    if (!rest || b.len < 24 * n) { PyBuffer_Release(&b); Py_XDECREF(rest); if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "gather3: destination too small"); return NULL; }
    double* out = (double*)b.buf;
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject* val = PyDict_GetItemWithError(param_dict, PyList_GET_ITEM(keys, i));
        int done = 0;
        Py_buffer v;
#ifndef PS_LOWER_NO_NUMPY
        if (val && PyArray_CheckExact(val)) {
            PyArrayObject* a = (PyArrayObject*)val;
            if (PyArray_TYPE(a) == NPY_DOUBLE && PyArray_ISNOTSWAPPED(a) && PyArray_ISALIGNED(a) && PyArray_SIZE(a) == 3 && PyArray_IS_C_CONTIGUOUS(a)) {
                memcpy(out + 3 * i, PyArray_DATA(a), 24); done = 1;
            }
        }
        if (!done)
#endif
        if (val && PyObject_GetBuffer(val, &v, PyBUF_FORMAT | PyBUF_C_CONTIGUOUS) == 0) {
            if (is_f64x3(&v)) { memcpy(out + 3 * i, v.buf, 24); done = 1; }
            PyBuffer_Release(&v);
        }
        PyErr_Clear();
        if (!done) { PyObject* ix = PyLong_FromSsize_t(i); if (!ix || PyList_Append(rest, ix)) { Py_XDECREF(ix); Py_DECREF(rest); PyBuffer_Release(&b); return NULL; } Py_DECREF(ix); }
    }
    PyBuffer_Release(&b);
    return rest;
}

/* scatter3(param_dict, keys: list, src: (n, 3) float64) -> list of the indices NOT written (value is not a writable array of three
 * contiguous doubles): val[...] = src[i] for the others */
static PyObject* scatter3(PyObject* self, PyObject* args) {
    PyObject *param_dict, *keys, *src;
    if (!PyArg_ParseTuple(args, "O!O!O", &PyDict_Type, &param_dict, &PyList_Type, &keys, &src)) return NULL;
    Py_buffer b;
    if (PyObject_GetBuffer(src, &b, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT)) return NULL;
    const Py_ssize_t n = PyList_GET_SIZE(keys);
    PyObject* rest = PyList_New(0);
    if (!rest || b.len < 24 * n || b.itemsize != 8) { PyBuffer_Release(&b); Py_XDECREF(rest); if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "scatter3: source too small"); return NULL; }
    const double* in = (const double*)b.buf;
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject* val = PyDict_GetItemWithError(param_dict, PyList_GET_ITEM(keys, i));
        int done = 0;
        Py_buffer v;
#ifndef PS_LOWER_NO_NUMPY
        if (val && PyArray_CheckExact(val)) {
            PyArrayObject* a = (PyArrayObject*)val;
            if (PyArray_TYPE(a) == NPY_DOUBLE && PyArray_ISNOTSWAPPED(a) && PyArray_ISALIGNED(a) && PyArray_SIZE(a) == 3 && PyArray_IS_C_CONTIGUOUS(a) &&
                PyArray_ISWRITEABLE(a)) {
                memcpy(PyArray_DATA(a), in + 3 * i, 24); done = 1;
            }
        }
        if (!done)
#endif
        if (val && PyObject_GetBuffer(val, &v, PyBUF_WRITABLE | PyBUF_FORMAT | PyBUF_C_CONTIGUOUS) == 0) {
            if (is_f64x3(&v)) { memcpy(v.buf, in + 3 * i, 24); done = 1; }
            PyBuffer_Release(&v);
        }
        PyErr_Clear();
        if (!done) { PyObject* ix = PyLong_FromSsize_t(i); if (!ix || PyList_Append(rest, ix)) { Py_XDECREF(ix); Py_DECREF(rest); PyBuffer_Release(&b); return NULL; } Py_DECREF(ix); }
    }
    PyBuffer_Release(&b);
    return rest;
}

static PyMethodDef methods[] = {
    {"classify", classify, METH_VARARGS, "split a parameter dictionary's keys into poses and 3-vector landmarks"},
    {"gather3", gather3, METH_VARARGS, "copy 3-vector parameters into the rows of an (n, 3) array; returns the indices left to the caller"},
    {"scatter3", scatter3, METH_VARARGS, "write the rows of an (n, 3) array into 3-vector parameters in place; returns the indices left to the caller"},
    {"walk", walk, METH_VARARGS, "take a run of consecutive reprojection blocks into the column arrays; returns (next block, count)"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_lower_fast", NULL, -1, methods};

PyMODINIT_FUNC PyInit__lower_fast(void) {
    s_KIND = PyUnicode_InternFromString("KIND"); s_camera = PyUnicode_InternFromString("camera");
    s_stiffness = PyUnicode_InternFromString("stiffness"); s_obs = PyUnicode_InternFromString("obs");
    s_CAMERA_ID = PyUnicode_InternFromString("CAMERA_ID");
#ifndef PS_LOWER_NO_NUMPY
    import_array();
#endif
    return PyModule_Create(&moddef);
}
