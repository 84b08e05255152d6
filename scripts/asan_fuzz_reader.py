"""Memory-safety fuzz of the native reader (csrc_host/tfrecord_reader.cpp) under AddressSanitizer + UBSan:
bit-flipped / truncated / random Example payloads and corrupted TFRecord framing must be rejected or decoded,
never read out of bounds.

    g++ -O1 -g -std=c++17 -fPIC -shared -pthread -fsanitize=address,undefined -fno-omit-frame-pointer \
        -Iinclude recalgorithm_amd/csrc_host/tfrecord_reader.cpp -o /tmp/librecalgo_host_asan.so
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python scripts/asan_fuzz_reader.py

Round 1: clean (150 mutated files + framing corruption, epochs + shuffle on the valid file).
Round 4: second phase through the asynchronous pipeline (recalgo_pipeline_*: single-valued columns only), also run under
ThreadSanitizer (-fsanitize=thread, LD_PRELOAD libtsan.so, RECALGO_READER_THREADS=6): clean.
Round 5: after the canonical-encoding fast paths, the huge-page vocabulary tables, the one-multiplication key hash and the
multi-valued-column fix of the pipeline: both phases clean under ASan + UBSan and under TSan (the bag column of this script's
dataset is what exposed the pipeline's -1 instead of -2)."""
import sys, os, tempfile, numpy as np
sys.path.insert(0,'/root/repo')
from recalgorithm_amd.io import native
native.LIB_PATH='/tmp/librecalgo_host_asan.so'
native.load('/tmp/librecalgo_host_asan.so')
from recalgorithm_amd import feature_column as fc
from recalgorithm_amd.io import synth, tfrecord as T
d=tempfile.mkdtemp()
spec = synth.SynthSpec(n_fields=6, max_vocab=300, seed=5, oov_frac=0.1, with_dense=True, with_history=True, with_tags=True)
vd=d+"/vocabulary/"; synth.write_vocabularies(spec, vd)
path=d+"/ex.tfrecord"; synth.write_tfrecord(spec, path, 64, chunk=64)
cols=[fc.embedding_column(fc.categorical_column_with_vocabulary_file(nm, vd+nm+".txt"), 8) for nm in spec.names]
cols+=[fc.embedding_column(fc.categorical_column_with_vocabulary_file("manual_tag_list", vd+"manual_tag_id.txt"), 8)]
labels=[fc.numeric_column("read_comment", default_value=0.0)]
n_ok=n_err=0
for _ in native.NativeDataset(path, cols+labels, ["read_comment"], 16, num_epochs=2, shuffle_buffer_size=7): n_ok+=1
recs=list(T.read_records(path)); rng=np.random.default_rng(0)
for case in range(150):
    muts=[]
    for r in recs[:16]:
        b=bytearray(r); k=case%3
        if k==0 and b:
            for _ in range(4): b[int(rng.integers(0,len(b)))]^=int(rng.integers(1,256))
        elif k==1: b=b[:int(rng.integers(0,len(b)+1))]
        else: b=bytearray(rng.integers(0,256,int(rng.integers(0,300)),dtype=np.uint8).tobytes())
        muts.append(bytes(b))
    p=d+f"/f{case}.tfrecord"; T.write_records(p,muts)
    try:
        for _ in native.NativeDataset(p, cols+labels, ["read_comment"], 8): pass
        n_ok+=1
    except (IOError, ValueError): n_err+=1
    # framing corruption too
    raw=bytearray(open(p,'rb').read())
    if raw:
        raw[int(rng.integers(0,len(raw)))]^=0xFF
        open(p,'wb').write(bytes(raw[:int(rng.integers(1,len(raw)+1))]))
        try:
            for _ in native.NativeDataset(p, cols+labels, ["read_comment"], 8, verify_crc=bool(case%2)): pass
        except (IOError, ValueError): n_err+=1
print("asan fuzz done", n_ok, n_err)
# ---- phase 2: the asynchronous pipeline (every column single-valued, so NativeDataset serves it from recalgo_pipeline_*) ----
spec2 = synth.SynthSpec(n_fields=6, max_vocab=300, seed=6, oov_frac=0.1)
vd2=d+"/vocabulary2/"; synth.write_vocabularies(spec2, vd2)
path2=d+"/ex2.tfrecord"; synth.write_tfrecord(spec2, path2, 200, chunk=64)
cols2=[fc.embedding_column(fc.categorical_column_with_vocabulary_file(nm, vd2+nm+".txt"), 8) for nm in spec2.names]
ds=native.NativeDataset(path2, cols2+labels, ["read_comment"], 16, num_epochs=3, shuffle_buffer_size=7)
assert ds._pipeline_columns() is not None
n_ok=n_err=0
for _ in ds: n_ok+=1
it=iter(native.NativeDataset(path2, cols2+labels, ["read_comment"], 16, num_epochs=-1)); [next(it) for _ in range(5)]; it.close()   # early close
recs=list(T.read_records(path2))
for case in range(150):
    muts=[]
    for r in recs[:40]:
        b=bytearray(r); k=case%3
        if k==0 and b:
            for _ in range(4): b[int(rng.integers(0,len(b)))]^=int(rng.integers(1,256))
        elif k==1: b=b[:int(rng.integers(0,len(b)+1))]
        else: b=bytearray(rng.integers(0,256,int(rng.integers(0,300)),dtype=np.uint8).tobytes())
        muts.append(bytes(b))
    p=d+f"/g{case}.tfrecord"; T.write_records(p,muts)
    try:
        for _ in native.NativeDataset(p, cols2+labels, ["read_comment"], 8): pass
        n_ok+=1
    except (IOError, ValueError): n_err+=1
    raw=bytearray(open(p,'rb').read())
    if raw:
        raw[int(rng.integers(0,len(raw)))]^=0xFF
        open(p,'wb').write(bytes(raw[:int(rng.integers(1,len(raw)+1))]))
        try:
            for _ in native.NativeDataset(p, cols2+labels, ["read_comment"], 8, verify_crc=bool(case%2)): pass
        except (IOError, ValueError): n_err+=1
print("pipeline fuzz done", n_ok, n_err)
