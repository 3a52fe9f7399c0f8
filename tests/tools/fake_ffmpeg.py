#!/usr/bin/env python3
"""Stand-in for the `ffmpeg` / `ffprobe` binaries (TEST TOOL: the image has neither; the reference bundles its own, a missing blob).

It speaks exactly the command lines backend/tools/video_io.py and backend/main.py issue -- nothing else -- so that the pipe
framing, the probe parsing, rotation handling, error propagation and the audio-mux call sequence of the product run against a
process on the other end of a pipe:

  ffprobe -v error -select_streams v:0 -count_packets -show_entries ... -of json FILE      -> the JSON ffprobe prints
  ffmpeg  -loglevel error -i FILE -f rawvideo -pix_fmt bgr24 -                          -> raw frames on stdout ("decoder")
  ffmpeg  -y -f rawvideo -vcodec rawvideo -s WxH -pix_fmt bgr24 -r FPS -i - ... OUT     -> frames from stdin into OUT ("encoder")

"Video files" are `FAKEVID1` containers: one JSON header line (w, h, fps, frames, rotation, known: whether the container knows its
frame count) followed by raw bgr24 frames.  A stored rotation of +-90 is applied on decode, as ffmpeg's autorotate does.
FAKE_FFMPEG_FAIL=encode makes the encoder exit 1 after three frames with a message on stderr; =exit makes it exit 3 at the end.
"""
import json
import os
import sys

import numpy as np

MAGIC = b"FAKEVID1\n"


def read_header(f):
    assert f.read(len(MAGIC)) == MAGIC, "not a FAKEVID1 file"
    return json.loads(f.readline())


def main():
    argv = sys.argv[1:]
    tool = os.environ.get("FAKE_TOOL") or os.path.basename(sys.argv[0])
    if tool.startswith("ffprobe"):
        with open(argv[-1], "rb") as f:
            h = read_header(f)
        st = {"width": h["w"], "height": h["h"], "r_frame_rate": h["fps"], "avg_frame_rate": h["fps"],
              "nb_read_packets": str(h["frames"]) if h.get("known", True) else "N/A", "nb_frames": "N/A"}
        num, den = h["fps"].split("/")
        if h.get("rotation"):
            st["side_data_list"] = [{"side_data_type": "Display Matrix", "rotation": h["rotation"]}]
        print(json.dumps({"streams": [st], "format": {"duration": str(h["frames"] * float(den) / float(num))}}))
        return 0
    if "-f" in argv and argv[argv.index("-i") + 1] == "-":                     # encoder
        w, h = (int(v) for v in argv[argv.index("-s") + 1].split("x"))
        fps = argv[argv.index("-r") + 1]
        out = argv[-1]
        fail = os.environ.get("FAKE_FFMPEG_FAIL")
        n = 0
        with open(out, "wb") as f:
            f.write(MAGIC)
            f.write(b" " * 200 + b"\n")
            while True:
                buf = sys.stdin.buffer.read(w * h * 3)
                if len(buf) < w * h * 3:
                    break
                f.write(buf)
                n += 1
                if fail == "encode" and n == 3:
                    sys.stderr.write("fake_ffmpeg: Error while encoding: out of tea\n")
                    return 1
            f.seek(len(MAGIC))
            head = json.dumps({"w": w, "h": h, "fps": f"{fps}/1" if "/" not in fps else fps, "frames": n}).encode()
            f.write(head + b" " * (200 - len(head)))
        if fail == "exit":
            sys.stderr.write("fake_ffmpeg: muxer said no\n")
            return 3
        return 0
    if "-i" in argv and argv[-1] == "-":                                     # decoder
        with open(argv[argv.index("-i") + 1], "rb") as f:
            h = read_header(f)
            for _ in range(h["frames"]):
                fr = np.frombuffer(f.read(h["w"] * h["h"] * 3), np.uint8).reshape(h["h"], h["w"], 3)
                rot = int(round(h.get("rotation", 0)))
                if rot % 360 in (90, -270):
                    fr = np.rot90(fr, 1)
                elif rot % 360 in (270, -90):
                    fr = np.rot90(fr, -1)
                sys.stdout.buffer.write(np.ascontiguousarray(fr).tobytes())
        return 0
    if "-c" in argv or "-map" in argv or "-vn" in argv:                       # the audio mux / extract calls of main.py: copy the video through
        outs = [a for a in argv if not a.startswith("-")]
        sys.stderr.write("fake_ffmpeg: no audio stream\n")
        return 1
    sys.stderr.write(f"fake_ffmpeg: unsupported command line {argv}\n")
    return 2


if __name__ == "__main__":
    sys.exit(main())
