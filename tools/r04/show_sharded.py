import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
if "error" in d:
    print(d); sys.exit(0)
c = d["config"]
print("value", d["value"], "ms", d["ms_per_step"], "| merge", c["merge"], "lanes", c["frames_in_flight"], "|", c["collectives_issued_by"][:30], "| note", c["comm_note"])
print("  via torch", c["Mevents_per_s_via_torch_distributed"], "enq", c["host_enqueue_us_per_frame_via_torch_distributed"], "| one at a time", c["Mevents_per_s_one_frame_at_a_time"],
      "| host enqueue us/frame", d["timing"]["host_enqueue_us_per_frame"])
print("  parity", d["parity"])
print("  kernels", d["roofline"]["avg_launch_us"], "coll", {k: v for k, v in d["collective_ms"].items() if k != "note"})
