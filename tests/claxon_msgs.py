"""Message / status ids parsed from include/claxon_hip.h (single source of truth)."""
import os
import re

_H = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "claxon_hip.h")


def _parse_enum(name):
    src = open(_H).read()
    body = re.search(r"typedef enum %s \{(.*?)\} %s;" % (name, name), src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out, nxt = {}, 0
    for item in body.split(","):
        item = item.strip()
        if not item:
            continue
        if "=" in item:
            k, v = [x.strip() for x in item.split("=")]
            nxt = int(v, 0)
        else:
            k = item
        out[k] = nxt
        nxt += 1
    return out


MSG = _parse_enum("clx_msg")
STATUS = _parse_enum("clx_status")
MSG_NAME = {v: k for k, v in MSG.items()}
