import random, statistics
random.seed(1)
def ray_len():
    # mixture: occluded shadow rays short, visible ones long, closest rays longest; mean ~16.4
    u=random.random()
    if u<0.40: return max(2,int(random.expovariate(1/7.0))+2)      # occluded: short
    if u<0.75: return max(6,int(random.gauss(22,7)))               # visible shadow
    return max(8,int(random.gauss(24,9)))                          # closest
def make_pool():
    return [ray_len() for _ in range(230)]
def run(steal, T_victim=32, T_thief=24, passes=300, shade=84):
    W=4
    t=0
    # wave state
    st=[]
    for w in range(W):
        st.append(dict(phase='shade', left=random.randint(0,shade), pool=[], lanes=[0]*64, src=[None]*64, pending=0, done_pass=0, busy=0, idle=0))
    total_lane_steps=0
    while min(s['done_pass'] for s in st)<passes:
        t+=1
        for w,s in enumerate(st):
            if s['phase']=='shade':
                s['left']-=1
                if s['left']<=0:
                    s['phase']='pass'; s['pool']=make_pool(); s['pool'].sort(reverse=False)  # random order (unsorted) -- shuffle
                    random.shuffle(s['pool']); s['lanes']=[0]*64; s['src']=[None]*64; s['pending']=0
                continue
            # pass: each lane with 0 remaining draws
            active=0
            for l in range(64):
                if s['lanes'][l]>0:
                    s['lanes'][l]-=1
                    if s['lanes'][l]==0 and s['src'][l] is not None:
                        st[s['src'][l]]['pending']-=1; s['src'][l]=None
                if s['lanes'][l]==0:
                    if s['pool']:
                        s['lanes'][l]=s['pool'].pop(); s['src'][l]=None
                    elif steal:
                        inflight=sum(1 for x in s['lanes'] if x>0)
                        if inflight>=T_thief:
                            for v in range(W):
                                if v!=w and st[v]['phase']=='pass' and len(st[v]['pool'])>=T_victim:
                                    s['lanes'][l]=st[v]['pool'].pop(); s['src'][l]=v; st[v]['pending']+=1
                                    break
                if s['lanes'][l]>0: active+=1
            if active==0 and not s['pool']:
                if s['pending']==0:
                    s['phase']='shade'; s['left']=shade; s['done_pass']+=1
                else:
                    s['idle']+=1   # waiting for thieves
            else:
                s['busy']+=1
    return t, sum(s['busy'] for s in st)/W/passes, sum(s['idle'] for s in st)/W/passes
for steal in (False, True):
    t,b,i=run(steal)
    print('steal',steal,'time',t,'pass iterations per pass',round(b,1),'victim wait per pass',round(i,2))
for tv,tt in ((16,16),(32,32),(48,24),(8,8)):
    t,b,i=run(True,tv,tt)
    print('T_victim',tv,'T_thief',tt,'time',t,'iters/pass',round(b,1),'wait',round(i,2))
