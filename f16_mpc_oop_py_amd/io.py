"""Trajectory writers: .npz for collated state-major trajectories and the comma-separated
time-history format of the reference's Simulink driver (Nguyen_m/runF16Sim.m:130-150; sample files C/ele_*.txt):

    time,npos,epos,alt,phi,theta,psi,vel,alpha,beta,p,q,r,nx,ny,nz,mach,qbar,ps,  + thrust,el,ail,rud

angles and rates in degrees as in those files, every value printed '%8.5f,'.  The six outputs nx..ps are the
Nlplant outputs (C/nlplant.c:443-450) and are evaluated on the GPU for the stored samples."""
import numpy as np
import torch


def save_npz(path, traj, dt, every=1, status=None, **meta):
    """traj: [T,18,B] tensor/array (state-major, as returned by F16Batch.rollout / dist.all_gather_trajectories)."""
    t = traj.detach().cpu().numpy() if isinstance(traj, torch.Tensor) else np.asarray(traj)
    extra = {} if status is None else {"status": status.detach().cpu().numpy() if isinstance(status, torch.Tensor) else status}
    np.savez_compressed(path, traj=t, time=(np.arange(t.shape[0]) + 1) * dt * every,
                        states=np.array(['npos', 'epos', 'h', 'phi', 'theta', 'psi', 'V', 'alpha', 'beta', 'p', 'q', 'r', 'T',
                                         'dh', 'da', 'dr', 'lf2', 'lf1']), **extra, **meta)


def save_csv(path, env, traj, aircraft=0, every=1, title="hifi DATA"):
    """One aircraft's time history in the runF16Sim.m text format."""
    T = traj.shape[0]
    xs = traj[:, :, aircraft]                                   # [T,18]
    from .env import F16Batch
    tmp = F16Batch(xs.detach().cpu().numpy(), xcg=env.xcg, fi_flag=env.fi_flag, dt=env.dt, device=env.device, context=env.ctx)
    out = tmp.nlplant(xs).cpu().numpy()                         # [T,18]: 12..17 = nx,ny,nz,mach,qbar,ps
    x = xs.detach().cpu().numpy()
    r2d = 180.0 / np.pi
    rows = np.column_stack([(np.arange(T) + 1) * env.dt * every, x[:, 0:3], x[:, 3:6] * r2d, x[:, 6], x[:, 7:9] * r2d,
                            x[:, 9:12] * r2d, out[:, 12:18], x[:, 12:16]])
    with open(path, "w") as f:
        f.write(f"% \n\t\t  {title}\n\n")
        f.write("\ntime,npos,epos,alt,phi,theta,psi,vel,alpha,beta,p,q,r,nx,ny,nz,mach,qbar,ps,\n\n")
        for row in rows:
            f.write("".join("%8.5f," % v for v in row) + "\n")
    return rows
