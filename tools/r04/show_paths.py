import json,sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"])
print("host_path", json.dumps(d.get("host_path"))[:1500])
ip = d.get("ingest_path", {})
print("ingest_path", json.dumps({k: v for k, v in ip.items() if k != "from_evt3_words"})[:2500])
