"""collect.sh step 4: fail unless traffic.json was measured on the library in the tree (sha256 stamp)."""
import hashlib, json, sys
t = json.load(open(sys.argv[1]))
sha = hashlib.sha256(open(sys.argv[2], "rb").read()).hexdigest()
if t.get("_lib_sha256") != sha:
    sys.exit(f"traffic.json ({t.get('_lib_sha256')}) is not from this libloco_hip.so ({sha})")
print("traffic.json matches the library:", sha[:16])
