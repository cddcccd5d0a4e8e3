import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/swift-qwen3-tts_amd')
from qwen3tts import synth, Qwen3TTSModel, GenerationRequest
d='/tmp/ckpt_tiny-b'; synth.write_checkpoint(d,'tiny-b')
def reqs(n):
    out=[]
    for r in range(n):
        pr=synth.synthetic_prompt(r,n_text=8+3*r,text_vocab=1000,im_start=1000,im_end=1001)
        out.append(GenerationRequest(pr['text_ids'],pr['target_token_count'],None,['aiden','vivian','eric'][r%3],['english','auto','chinese'][r%3]))
    return out
R=reqs(6)
ref=None
for lanes,graph in ((1,False),(1,True),(2,True),(3,True)):
    m=Qwen3TTSModel.from_pretrained(d,max_batch=6,max_frames=32,max_prompt=64,use_graph=graph,n_streams=lanes)
    res=m.generate_batch(R,temperature=0.9,seed=7,force_frames=10)
    codes=np.stack([r.codes for r in res]); pcm=np.stack([r.audio for r in res])
    if ref is None: ref=(codes,pcm)
    print('lanes',lanes,'graph',graph,'codes equal',np.array_equal(codes,ref[0]),'pcm maxdiff',np.abs(pcm-ref[1]).max())
    # batch-1 equivalence for row 3
    if lanes==1 and graph:
        m1=Qwen3TTSModel.from_pretrained(d,max_batch=1,max_frames=32,max_prompt=64)
        r3=m1.generate_batch([R[3]],temperature=0.0,force_frames=10)[0]
        rb=m.generate_batch(R,temperature=0.0,force_frames=10)[3]
        print('row independence greedy', np.array_equal(r3.codes, rb.codes), np.abs(r3.audio-rb.audio).max())
    m.close()
