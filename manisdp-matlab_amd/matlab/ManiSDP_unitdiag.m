% Drop-in for src/primal/ManiSDP_unitdiag.m:  Min <C,X> s.t. A(X) = b, X >= 0, diag(X) = 1.
% Same signature / options / outputs; trustregions() -> manisdp_mex('rtr', ...).
function [X, obj, data] = ManiSDP_unitdiag(At, b, c, K, options)

n = K.s;
if ~isfield(options,'p0'); options.p0 = 2; end
if ~isfield(options,'AL_maxiter'); options.AL_maxiter = 300; end
if ~isfield(options,'gama'); options.gama = 2; end
if ~isfield(options,'sigma0'); options.sigma0 = 1e-3; end
if ~isfield(options,'sigma_min'); options.sigma_min = 1e-2; end
if ~isfield(options,'sigma_max'); options.sigma_max = 1e7; end
if ~isfield(options,'tol'); options.tol = 1e-8; end
if ~isfield(options,'theta'); options.theta = 1e-3; end
if ~isfield(options,'delta'); options.delta = 8; end
if ~isfield(options,'alpha'); options.alpha = 0.1; end
if ~isfield(options,'tolgradnorm'); options.tolgradnorm = 1e-8; end
if ~isfield(options,'TR_maxinner'); options.TR_maxinner = 20; end
if ~isfield(options,'TR_maxiter'); options.TR_maxiter = 4; end
if ~isfield(options,'tau1'); options.tau1 = 1; end
if ~isfield(options,'tau2'); options.tau2 = 1; end
if ~isfield(options,'line_search'); options.line_search = 0; end

fprintf('ManiSDP is starting...\n');
fprintf('SDP size: n = %i, m = %i\n', n, size(b,1));

A = At';
h = manisdp_mex('create_unitdiag', At, b, c, n);
cleanup = onCleanup(@() manisdp_mex('destroy', h));
p = options.p0;
sigma = options.sigma0;
gama = options.gama;
y = zeros(length(b), 1);
normb = 1 + norm(b);
Y = randn(p, n);  Y = Y./sqrt(sum(Y.^2, 1));
U = [];
fac_size = [];
opts.maxinner = options.TR_maxinner;
opts.maxiter = options.TR_maxiter;
opts.tolgradnorm = options.tolgradnorm;

data.status = 0;
timespend = tic;
for iter = 1:options.AL_maxiter
    fac_size = [fac_size; p];
    manisdp_mex('set_multipliers', h, y, sigma);
    if ~isempty(U)
        Y = line_search(Y, U);
    end
    [Y, info] = manisdp_mex('rtr', h, Y, opts);
    gradnorm = info.gradnorm;
    X = Y'*Y;
    x = X(:);
    obj = c'*x;
    Axb = A*x - b;
    pinf = norm(Axb)/normb;
    y = y - sigma*Axb;
    eS = reshape(c - At*y, n, n);
    z = sum(X.*eS);
    S = eS - diag(z);
    [vS, dS] = eig(full(S), 'vector');
    dinf = max(0, -dS(1))/(1+dS(end));
    by = b'*y + sum(z);
    gap = abs(obj-by)/(abs(by)+abs(obj)+1);
    [~, D, V] = svd(Y, 'econ');
    e = diag(D);
    r = sum(e >= options.theta*e(1));
    fprintf('Iter %d, obj:%0.8f, gap:%0.1e, pinf:%0.1e, dinf:%0.1e, gradnorm:%0.1e, r:%d, p:%d, sigma:%0.3f, time:%0.2fs\n', ...
             iter,    obj,       gap,       pinf,       dinf,       gradnorm,       r,    p,    sigma,       toc(timespend));
    eta = max([gap, pinf, dinf]);
    if eta < options.tol
        fprintf('Optimality is reached!\n');
        break;
    end
    if mod(iter, 50) == 0
        if iter > 100 && gap > gap0 && pinf > pinf0 && dinf > dinf0
            data.status = 2;
            fprintf('Slow progress!\n');
            break;
        else
            gap0 = gap; pinf0 = pinf; dinf0 = dinf;
        end
    end
    if r <= p - 1
        Y = V(:,1:r)'.*e(1:r);
        p = r;
    end
    nne = max(min(sum(dS < 0), options.delta), 1);
    if options.line_search == 1
       U = [zeros(p, n); vS(:,1:nne)'];
    end
    p = p + nne;
    if options.line_search == 1
       Y = [Y; zeros(nne,n)];
    else
       Y = [Y; options.alpha*vS(:,1:nne)'];
       Y = Y./sqrt(sum(Y.^2));
    end
    if pinf < options.tau1*gradnorm
          sigma = max(sigma/gama, options.sigma_min);
    elseif pinf > options.tau2*gradnorm
          sigma = min(sigma*gama, options.sigma_max);
    end
end
data.X = X; data.y = y; data.S = S; data.z = z; data.gap = gap; data.pinf = pinf; data.dinf = dinf;
data.gradnorm = gradnorm; data.time = toc(timespend); data.fac_size = fac_size;
if data.status == 0 && eta > options.tol
    data.status = 1;
    fprintf('Iteration maximum is reached!\n');
end
fprintf('ManiSDP: optimum = %0.8f, time = %0.2fs\n', obj, toc(timespend));

    function nY = line_search(Y, U)
         alpha = 1;
         cost0 = manisdp_mex('linesearch_cost', h, Y, U, 0);
         i = 1;
         nY = Y + alpha*U;  nY = nY./sqrt(sum(nY.^2));
         while i <= 15 && manisdp_mex('linesearch_cost', h, Y, U, alpha) - cost0 > -1e-3
              alpha = 0.8*alpha;
              nY = Y + alpha*U;  nY = nY./sqrt(sum(nY.^2));
              i = i + 1;
         end
    end
end
