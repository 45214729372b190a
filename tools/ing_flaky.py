import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import ingest_oracle as IO, xmaps_oracle as O
from x_maps_amd import XMapsEngine, evt3, synthetic as S
from x_maps_amd.ingest import DeviceIngest
import test_gpu_ingest as TI
cfg = S.C_TINY
tb = S.make_tables(cfg)
stream = TI._tiny_stream(14, seed=7)
pks = [pk for pk in TI._packets(stream, int(1e6 / 60 / 4)) if len(pk)]
chunks = [evt3.encode_evt3(pk) for pk in pks]
tf = IO.TriggerFinderOracle(60)
hd = evt3.Evt3Decoder()
for c in chunks:
    tf.process_events(IO.polarity_filter(hd.decode(c)))
want = []
for evs in tf.frames:
    x, y, t, _ = S.to_soa(evs)
    want.append(int(O.process_ev_frame(tb, x.astype(np.int64), y.astype(np.int64), t)["mask"].sum()))
bad = {"evt3": 0, "records": 0}
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    host_dec = evt3.Evt3Decoder()
    with XMapsEngine(tb) as e1, XMapsEngine(tb) as e2:  # two engines and two ingests alive at once, as in the test
        ing1 = DeviceIngest(e1, 60, max_packet_events=8192, capacity_events=65536)
        ing2 = DeviceIngest(e2, 60, max_packet_events=8192, capacity_events=65536)
        out1, out2 = [], []
        with evt3.DeviceEvt3Decoder(e1, max_words=max(len(c) for c in chunks)) as dec:
            for words in chunks:
                dec.push(ing1, words)
                ing2.push(host_dec.decode(words))
                out1 += ing1.poll()
                out2 += ing2.poll()
            ing1.flush(); ing2.flush()
            out1 += ing1.poll(); out2 += ing2.poll()
        for mode, out in (("evt3", out1), ("records", out2)):
            got = [f.n_inliers for f in out]
            if got != want:
                bad[mode] += 1
                print(rep, mode, "got", got, "want", want, flush=True)
        ing1.close(); ing2.close()
print("failures", bad)
